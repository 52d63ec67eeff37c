"""Developer tool for the SLP question (Makefile target `slp`): bitwise re-run check of the fused head in every 16-bit mode with the
library given in CIAOSR_HIP_LIB.  Prints, per mode, how many of the runs differ from the first and where.
    CIAOSR_HIP_LIB=ciaosr_amd/csrc/libciaosr_hip_slp_head_fused_bf16.so python tools/slp_bisect.py [runs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                        # noqa: E402
from ciaosr_amd import hip_ops                      # noqa: E402
from ciaosr_amd.coords import make_coord, make_cell  # noqa: E402
from tests.helpers import SQRT6, randn, seeded_head  # noqa: E402
from tests.test_hip_parity import _my_generator     # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device('cuda:0')
g = _my_generator(64, (256,) * 4, seeded_head(64, 0, head_gain=SQRT6), dev, eval_bsize=30000)
feat = (randn((1, 64, 64, 64), 7) * 0.3).to(dev)
coord, cell = make_coord((256, 256)).unsqueeze(0).to(dev), make_cell((256, 256)).unsqueeze(0).to(dev)
x = (randn((1, 3, 64, 64), 14) * 0.3).to(dev)
lib = os.path.basename(os.environ.get('CIAOSR_HIP_LIB', 'libciaosr_hip.so'))
for name, opt in (('fp32', hip_ops.Options('fp32')), ('bf16', hip_ops.Options('bf16')), ('bf16-single', hip_ops.Options('bf16', bf16_single=1)),
                  ('f16', hip_ops.Options('f16')), ('f16-pairs', hip_ops.Options('f16-pairs')), ('f16x3', hip_ops.Options('f16x3')),
                  ('f16-wide', hip_ops.Options('f16', head_route=8))):
    outs = [g._predict([feat], coord, cell, 30000, x, opt).clone() for _ in range(runs)]
    bad = []
    for o in outs[1:]:
        ne = (outs[0] != o)
        if ne.any():
            q = ne.any(-1)[0].nonzero()[:, 0]
            bad.append((int(q.numel()), sorted(set((q % 128).tolist()))[:8], ne[0].any(0).tolist()))
    print(f'{lib:44s} {name:12s} differing runs {len(bad)}/{runs - 1}' + (f'  e.g. {bad[0][0]} queries, rows-in-wg {bad[0][1]}, channels {bad[0][2]}' if bad else ''), flush=True)
