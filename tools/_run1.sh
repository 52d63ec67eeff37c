python - <<'PY'
import sys; sys.path.insert(0, '.')
import torch
from bench import rdn_ciaosr
from ciaosr_amd import hip_ops
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
dev = torch.device('cuda:0')
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32)); seeded_init_(model, seed=0, gain=1.0); model = model.to(dev)
lq = synthetic_pair(339, 510, 4)[0].to(dev)
for _ in range(2): model.restore(lq)
torch.cuda.synchronize()
with hip_ops.profile():
    model.restore(lq); torch.cuda.synchronize()
r = hip_ops.profile.results()
for k in sorted(r, key=lambda k: -r[k]['total_ms'])[:40]:
    if 'pack' in k or 'cast' in k: print(k, r[k])
print({k: v for k, v in r.items() if 'pack' in k})
PY
