"""Developer lab: per-kernel time of one C3 tile (192x192 LR -> 768x768) under option variants, one process, one GPU.
   python tools/kernel_lab.py [--quick] [variant ...]     variants: name=precision[,field=value...]
   --quick skips the tile-stream timings; CIAOSR_HIP_LIB=<path> selects another build of the library (A/B of kernel variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import rdn_ciaosr, time_steps
from ciaosr_amd import hip_ops
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair

dev = torch.device('cuda:0')
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, seed=0, gain=1.0)
model = model.to(dev)
lq = synthetic_pair(192, 192, 4)[0].to(dev)
quick = '--quick' in sys.argv
variants = [a for a in sys.argv[1:] if a != '--quick'] or ['fp32=fp32', 'bf16=bf16', 'bf16_single=bf16,bf16_single=1']
ref = None
for v in variants:
    name, spec = v.split('=', 1)
    parts = spec.split(',')
    opt = hip_ops.Options(parts[0], **{k: int(x) for k, x in (p.split('=') for p in parts[1:])})
    out = model.restore(lq, options=opt)
    ms = time_steps(lambda: model.restore(lq, options=opt), 5, dev)
    with hip_ops.profile():
        model.restore(lq, options=opt)
        torch.cuda.synchronize()
    prof = hip_ops.profile.results()
    top = sorted(prof.items(), key=lambda kv: -kv[1]['total_ms'])[:16]
    if ref is None:
        ref = out
    d = (out - ref).abs()
    print(f'{name:12s} {ms:8.3f} ms/tile  max|d vs first| {d.max().item():.2e} rms {d.pow(2).mean().sqrt().item():.2e}  ' +
          ' '.join(f'{k}={x["total_ms"]:.2f}' for k, x in top), flush=True)

if quick:
    sys.exit(0)
# tile-stream concurrency on the 6-tile image
lq6 = synthetic_pair(339, 510, 4)[0].to(dev)
for prec in ('fp32', 'bf16', 'f16'):
    for ns in (1, 2, 3):
        model.test_cfg['tile_streams'] = ns
        model.restore(lq6, options=prec)
        ms = time_steps(lambda: model.restore(lq6, options=prec), 3, dev)
        print(f'c3s (6 tiles) {prec} tile_streams={ns}: {ms:8.2f} ms', flush=True)
