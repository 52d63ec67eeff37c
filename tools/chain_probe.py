"""Developer probe: phase stamps of the SECOND pass of every workgroup of head_kv_chain_kernel (f16) on one tile (needs `make -C
ciaosr_amd/csrc probe`).   CIAOSR_HIP_LIB=ciaosr_amd/csrc/libciaosr_hip_probe.so python tools/chain_probe.py [192] [f16|f16-pairs]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rdn_ciaosr                              # noqa: E402
from ciaosr_amd import _lib, hip_ops                      # noqa: E402
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair   # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 192
mode = sys.argv[2] if len(sys.argv) > 2 else 'f16'
dev = torch.device('cuda')
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, 0)
model = model.to(dev)
lq = synthetic_pair(size, size, 4)[0].to(dev)
opt = hip_ops.Options(mode)
for _ in range(3):
    model.restore(lq, options=opt)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * (256 * 16))()
lib.ciaosr_debug_probe_chain_read.restype = C.c_int
assert lib.ciaosr_debug_probe_chain_read(buf, 256 * 16) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16).astype(np.int64)
names = ['index math', 'build rows k', 'k hidden x3', 'logit + softmax', 'build rows v', 'v hidden x3', 'v out + epilogue']
d = a[:, 1:8] - a[:, 0:7]
tot = a[:, 7] - a[:, 0]
pairs = 2 if mode != 'f16' else 1
print(f'pass 1 of 256 workgroups; pass avg {tot.mean():.0f} cycles (min {tot.min()}, max {tot.max()})')
for i, nm in enumerate(names):
    print(f'  {nm:20s} {d[:, i].mean():9.0f} cycles avg  ({100 * d[:, i].mean() / tot.mean():5.1f} %)')
mf = 32 * 2 * 16 * pairs
print(f'  MFMA issue of one wave: hidden x3 {24 * mf}, v out (20 units) {20 * mf}, pass {68 * mf} ({100 * 68 * mf / tot.mean():.1f} % of the pass)')
