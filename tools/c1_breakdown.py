"""C1 (EDSR-baseline CiaoSR x2, LR 48x48 -> 96x96: the reference's own CPU-runnable case) step time and breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ciaosr_amd import hip_ops, build_model
from ciaosr_amd.config import Config
from ciaosr_amd.coords import make_coord, make_cell
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair

cfg = Config.fromfile(os.path.join(os.path.dirname(__file__), '..', 'configs', '001_localimplicitsr_edsr_div2k_g1_c64b16_1000k_unfold_lec_mulwkv_res_nonlocal.py'))
model = build_model(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
seeded_init_(model, seed=0, gain=1.0)
dev = torch.device('cuda')
model = model.to(dev).eval()
lq, _ = synthetic_pair(48, 48, 2)
lq = lq.to(dev)
coord, cell = make_coord((96, 96)).unsqueeze(0).to(dev), make_cell((96, 96)).unsqueeze(0).to(dev)
model.test_cfg['tile'] = None
for _ in range(3):
    out = model.restore(lq, coord, cell)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = model.restore(lq, coord, cell)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
print(f'C1 fp32: {ms:.3f} ms/img, {96 * 96 / 1e6 / (ms * 1e-3):.3f} HR Mpix/s, out {tuple(out.shape)}')
with hip_ops.profile():
    model.restore(lq, coord, cell)
    torch.cuda.synchronize()
for k, v in sorted(hip_ops.profile.results().items(), key=lambda kv: -kv[1]['total_ms'])[:10]:
    print(f'  {k:28s} {v["total_ms"]:8.3f} ms  x{v["launches"]}')
