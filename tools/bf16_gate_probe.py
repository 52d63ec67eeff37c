"""Developer probe: PSNR-delta-vs-GT of the bf16 mode on the reference vectors, by which kernels run in bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.helpers import load_golden, SQRT6
from tests.test_hip_parity import _restorer, _t
from ciaosr_amd import hip_ops
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
from ciaosr_amd.metrics import psnr_tensors

dev = torch.device('cuda:0')
fx = load_golden('e2e_rdn_x4_48')
model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=SQRT6)
model = model.to(dev)
lq = _t(fx['lq']).to(dev)
ref = _t(fx['out'])
_, gt = synthetic_pair(48, 48, 4)
p_ref = psnr_tensors(ref, gt, crop_border=4)
O = hip_ops.Options
for name, opt in (('fp32', O('fp32')), ('bf16 natural (head only at 48x48)', O('bf16')),
                  ('bf16 + dense forced', O('bf16', dense_min_tiles=1)),
                  ('bf16 + csa forced', O('bf16', csa_composed_min=1)),
                  ('bf16 all forced', O('bf16', dense_min_tiles=1, csa_composed_min=1))):
    out = model.restore(lq, options=opt).cpu()
    d = (out - ref)
    print(f'{name:40s} max|d| {d.abs().max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e} '
          f'PSNR delta vs GT {abs(psnr_tensors(out, gt, crop_border=4) - p_ref):.5f} dB')
fx = load_golden('e2e_rdn_x4_tile192')
model = _restorer('rdn', 4, dev, dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, seed=int(fx['weight_seed']), gain=float(fx['gain']), head_gain=SQRT6)
model = model.to(dev)
lq, gt = synthetic_pair(192, 192, 4)
for name, opt in (('tile192 fp32', O('fp32')), ('tile192 bf16 (hi+lo weights)', O('bf16')), ('tile192 bf16 single', O('bf16', bf16_single=1)),
                  ('tile192 bf16 pairs, trunk fp32', O('bf16', dense_min_tiles=-1)),
                  ('tile192 bf16 pairs, head only', O('bf16', dense_min_tiles=-1, csa_composed_min=-1))):
    out = model.restore(lq.to(dev), options=opt).cpu()
    d = out[..., ::4, ::4] - _t(fx['out_s4'])
    print(f'{name:40s} (gain {float(fx["gain"])}) max|d| {d.abs().max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e} '
          f'PSNR delta vs GT {abs(psnr_tensors(out, gt, crop_border=4) - float(fx["psnr_ref_gt"])):.5f} dB')
