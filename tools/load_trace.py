"""Model-load path under `rocprofv3 --kernel-trace`: builds RDN-CiaoSR, runs prepare() + one 48x48 restore in fp32 (and, with an argument,
in that 16-bit mode too), prints prepare() wall time.  tools/load_trace_summary.py counts the kernels of the trace by name:
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/load -o t -- python3 tools/load_trace.py [f16]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rdn_ciaosr                                     # noqa: E402
from ciaosr_amd import hip_ops                                   # noqa: E402
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair   # noqa: E402

dev = torch.device('cuda', 0)
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, seed=0, gain=1.0)
model = model.to(dev)
lq = synthetic_pair(48, 48, 4)[0].to(dev)
for prec in ['fp32'] + sys.argv[1:]:
    opt = hip_ops.Options(prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.prepare(opt)
    torch.cuda.synchronize()
    print(f'prepare({prec}): {time.perf_counter() - t0:.3f} s', flush=True)
    out = model.restore(lq, options=opt)
    torch.cuda.synchronize()
    print(f'restore({prec}): out std {float(out.std()):.4f}', flush=True)
