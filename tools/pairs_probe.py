"""Developer probe: which stage of the f16-pairs mode carries its error on the full C3 tile (tests/golden/e2e_rdn_x4_tile192.npz)?
   Runs the trunk and the head (cs_attn + MLPs) in different precisions and prints max / rms |delta| against the reference subset."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import rdn_ciaosr
from ciaosr_amd import hip_ops
from ciaosr_amd.coords import make_coord, make_cell
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair

dev = torch.device('cuda:0')
fx = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'e2e_rdn_x4_tile192.npz'))
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=6 ** 0.5)
model = model.to(dev)
g = model.generator
lq, gt = synthetic_pair(192, 192, 4)
x = model.normalize(lq.to(dev))
ref = torch.from_numpy(fx['out_s4'])
coord, cell = make_coord((768, 768)).to(dev), make_cell((768, 768)).to(dev)
names = {'fp32': hip_ops.Options('fp32'), 'f16': hip_ops.Options('f16'), 'f16p': hip_ops.Options('f16-pairs')}
for tn in ('fp32', 'f16p', 'f16'):
    feat = g._encoder_hip.forward_hwc(x[0], names[tn])
    for hn in ('fp32', 'f16p', 'f16'):
        rgb = g._head.forward(None, x[0], coord, cell, 30000, feature_hwc=feat, options=names[hn])
        out = hip_ops.denorm_clamp(rgb.contiguous(), 768, 768, model.rgb_mean, model.rgb_std).unsqueeze(0)
        d = (out[..., ::4, ::4].cpu() - ref).abs()
        print(f'trunk {tn:5s} head {hn:5s}: max |d| {d.max().item():.3e}  rms {d.double().pow(2).mean().sqrt().item():.3e}', flush=True)
