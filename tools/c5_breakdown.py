"""C5 (SwinIR-CiaoSR x3.3, LR 48x48 -> 158x158) step time and per-kernel breakdown (developer tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ciaosr_amd import hip_ops
from ciaosr_amd.config import Config
from ciaosr_amd import build_model
from ciaosr_amd.coords import make_coord, make_cell
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
cfg = Config.fromfile(os.path.join(os.path.dirname(__file__), '..', 'configs', '001_localimplicitsr_swinir_div2k_g1_c64b16_1000k_unfold_lec_mulwkv_res_nonlocal.py'))
model = build_model(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
seeded_init_(model, seed=0, gain=1.0)
dev = torch.device('cuda')
model = model.to(dev).eval()
lq, _ = synthetic_pair(48, 48, 3)
lq = lq.to(dev)
ht = wt = 158
coord, cell = make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev)
model.test_cfg['tile'] = None
model.test_cfg['precision'] = prec
for _ in range(3):
    out = model.restore(lq, coord, cell)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = model.restore(lq, coord, cell)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
print(f'C5 {prec}: {ms:.3f} ms/img, {ht * wt / 1e6 / (ms * 1e-3):.3f} HR Mpix/s, out {tuple(out.shape)}')
x = model.normalize(lq)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    f = model.generator.gen_feature(x)
torch.cuda.synchronize()
print(f'  gen_feature (SwinIR trunk, HIP: swinir.hip): {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms')
with hip_ops.profile():
    model.restore(lq, coord, cell)
    torch.cuda.synchronize()
for k, v in sorted(hip_ops.profile.results().items(), key=lambda kv: -kv[1]['total_ms'])[:12]:
    print(f'  {k:28s} {v["total_ms"]:8.3f} ms  x{v["launches"]}')
# hipGraph replay of the whole step (the PyTorch-ROCm trunk is ~1000 tiny launches: CPU-launch bound in eager mode)
try:
    run = model.graphed_restore(lq, coord, cell)
    ref = model.restore(lq, coord, cell)
    g = run()
    torch.cuda.synchronize()
    print('  graph replay equals eager:', torch.equal(g, ref))
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f'  hipGraph replay: {ms:.3f} ms/img, {ht * wt / 1e6 / (ms * 1e-3):.3f} HR Mpix/s')
except Exception as e:          # noqa: BLE001 - developer tool: report why the capture failed
    print('  graph capture failed:', repr(e)[:300])
