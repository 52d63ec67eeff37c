python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -n 6
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from helpers import randn, csattn_shapes
from test_hip_parity import _my_csattn
from ciaosr_amd import hip_ops
from ciaosr_amd.init_utils import seeded_state_dict
dev = torch.device('cuda:0')
P = seeded_state_dict(csattn_shapes(64, prefix=''), 11, 1.0)
att = _my_csattn(64, P, dev, prefix='')
for hw in ((256, 256), (300, 300)):
    x = randn((1, 64) + hw, 12).to(dev)
    ref = None
    for prec in ('fp32', 'f16', 'bf16'):
        try:
            y = att(x, options=hip_ops.Options(prec)).cpu()
            if ref is None: ref = y
            print(hw, prec, 'ok', (y - ref).abs().max().item())
        except Exception as e:
            print(hw, prec, 'FAILED', str(e)[:150])
PY
