"""Developer probe: which torch (aten) ops the product path itself launches per restore() -- every one of them is a small kernel between
the library's launches.  Groups aten::copy_ / clone / contiguous / cat / zeros calls by the innermost ciaosr_amd source line.
   python tools/aten_ops.py [f16|fp32] [lr_size]"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rdn_ciaosr                              # noqa: E402
from ciaosr_amd import hip_ops                            # noqa: E402
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair   # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'f16'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 192
dev = torch.device('cuda')
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, 0)
model = model.to(dev)
lq = synthetic_pair(size, size, 4)[0].to(dev)
opt = hip_ops.Options(mode)
for _ in range(2):
    model.restore(lq, options=opt)
torch.cuda.synchronize()

counts = collections.Counter()


class Spy(torch.utils._python_dispatch.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ('copy_', 'clone', '_to_copy', 'cat', 'stack', 'zeros', 'fill_', 'index', 'gather')):
            fr = [f for f in traceback.extract_stack() if 'ciaosr_amd' in f.filename]
            where = f'{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line}' if fr else '?'
            counts[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    model.restore(lq, options=opt)
torch.cuda.synchronize()
for (name, where), n in counts.most_common(40):
    print(f'{n:4d}  {name:28s} {where}')
