"""Host/device time breakdown of one C2 step (developer tool, run on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rdn_ciaosr
from ciaosr_amd import hip_ops
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
from ciaosr_amd.coords import make_coord, make_cell

dev = torch.device('cuda:0')
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, 0)
model = model.to(dev)
lq, _ = synthetic_pair(48, 48, 4)
lq = lq.to(dev)
g = model.generator
x = model.normalize(lq)
coord = make_coord((192, 192)).to(dev).contiguous()
cell = make_cell((192, 192)).to(dev).contiguous()


def timeit(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name:28s} enqueue {1e3 * (t1 - t0) / n:8.3f} ms   total {1e3 * (t2 - t0) / n:8.3f} ms')


feat = g._encoder_hip.forward_hwc(x[0])
timeit('normalize', lambda: model.normalize(lq))
timeit('encoder_hip', lambda: g._encoder_hip.forward_hwc(x[0]))
timeit('head', lambda: g._head.forward(None, x[0], coord, cell, 30000, feature_hwc=feat))
timeit('generator.forward', lambda: g(x, coord.unsqueeze(0), cell.unsqueeze(0), test_mode=True))
timeit('restore (full step)', lambda: model.restore(lq))
timeit('make_coord+cell to dev', lambda: (make_coord((192, 192)).to(dev), make_cell((192, 192)).to(dev)))
timeit('struct() enc', lambda: g._encoder_hip.struct())
timeit('struct() head', lambda: g._head.struct())

if os.environ.get('CPROFILE'):
    import cProfile, pstats
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        model.restore(lq)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
