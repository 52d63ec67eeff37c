"""Developer lab: the RDN trunk of a batch of C3 tiles (8 x 192x192) alone, per-kernel ms.  python tools/enc_lab.py [precision] [batch]
CIAOSR_HIP_LIB=<path> selects another build of the library (A/B of kernel variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import rdn_ciaosr, time_steps
from ciaosr_amd import hip_ops
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair

dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, seed=0, gain=1.0)
model = model.to(dev)
g = model.generator
enc = g._encoder_hip
x = synthetic_pair(192, 192, 4)[0].expand(B, -1, -1, -1).contiguous().to(dev)
opt = hip_ops.Options(prec)
out = enc.forward_hwc_batch(x, opt)
ms = time_steps(lambda: enc.forward_hwc_batch(x, opt), 5, dev)
with hip_ops.profile():
    enc.forward_hwc_batch(x, opt)
    torch.cuda.synchronize()
prof = hip_ops.profile.results()
top = sorted(prof.items(), key=lambda kv: -kv[1]['total_ms'])[:6]
print(f'{prec} trunk, batch {B}: {ms:8.3f} ms  ({ms / B:.3f} per tile)  ' + ' '.join(f'{k}={x_["total_ms"]:.2f}' for k, x_ in top), flush=True)
