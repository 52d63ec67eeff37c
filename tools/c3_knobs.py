import sys, time, torch
sys.path.insert(0, '/root/repo')
from bench import rdn_ciaosr, time_steps
from ciaosr_amd import hip_ops
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
dev = torch.device('cuda', 0)
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, seed=0, gain=1.0)
model = model.to(dev)
lq = synthetic_pair(1356, 2040, 4)[0].to(dev)
ref = None
for name, cfg in (('default', {}), ('encoder_ahead', dict(encoder_ahead=True)), ('tile_streams2', dict(tile_streams=2)), ('both', dict(encoder_ahead=True, tile_streams=2)), ('streams3', dict(tile_streams=3))):
    for k in ('encoder_ahead', 'tile_streams'):
        model.test_cfg.pop(k, None)
    model.test_cfg.update(cfg)
    out = model.restore(lq)
    if ref is None: ref = out.clone()
    t = time_steps(lambda: model.restore(lq), 3, dev)
    print(f'{name:14s} {t:8.1f} ms  bitwise {torch.equal(out, ref)}', flush=True)
    del out
