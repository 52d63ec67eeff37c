"""Developer probe: per-phase cycle stamps of head_kv_fused_kernel at C2 (needs `make -C ciaosr_amd/csrc probe`).
   CIAOSR_HIP_LIB=ciaosr_amd/csrc/libciaosr_hip_probe.so python tools/head_probe.py"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import rdn_ciaosr
from ciaosr_amd import _lib
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair

dev = torch.device('cuda')
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, 0)
model = model.to(dev)
import sys as _s
HW_ = int(_s.argv[1]) if len(_s.argv) > 1 else 48
lq, _ = synthetic_pair(HW_, HW_, 4)
lq = lq.to(dev)
MODE = _s.argv[2] if len(_s.argv) > 2 else 'fp32'          # fp32 | bf16 | bf16-single | f16x3
from ciaosr_amd import hip_ops
OPT = hip_ops.Options('fp32') if MODE == 'fp32' else (hip_ops.Options('f16x3') if MODE == 'f16x3' else hip_ops.Options('f16') if MODE == 'f16w' else hip_ops.Options('bf16', bf16_single=int(MODE == 'bf16-single')))
for _ in range(3):
    model.restore(lq, options=OPT)
torch.cuda.synchronize()
lib = _lib.load()
n = 4096
buf = (C.c_ulonglong * (4096 * 16))()
reader = lib.ciaosr_debug_probe_read if MODE == 'fp32' else (lib.ciaosr_debug_probe_x3_read if MODE in ('f16x3', 'f16w') else lib.ciaosr_debug_probe16_read)
reader.restype = C.c_int
assert reader(buf, 4096 * 16) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16)[:n].astype(np.int64)
names = ['index math', 'build rows k', 'k hidden x3', 'logit + softmax', 'build rows v', 'v hidden x3', 'v out + epilogue']
d = a[:, 1:8] - a[:, 0:7]
tot = a[:, 7] - a[:, 0]
print(f'{n} workgroups; lifetime avg {tot.mean():.0f} ticks (min {tot.min()}, max {tot.max()})')
for i, nm in enumerate(names):
    print(f'  {nm:20s} {d[:, i].mean():9.0f} ticks avg  ({100 * d[:, i].mean() / tot.mean():5.1f} %)')
if MODE in ('f16x3', 'f16w'):
    kl = [a[:, 10] - a[:, 2], a[:, 11] - a[:, 10], a[:, 12] - a[:, 11]]
    vl = [a[:, 13] - a[:, 5], a[:, 14] - a[:, 13], a[:, 15] - a[:, 14]]
    print('  k hidden layers 1..3: ' + ', '.join(f'{x.mean():.0f}' for x in kl) + '   v hidden layers 1..3: ' + ', '.join(f'{x.mean():.0f}' for x in vl))
    per_step = 12 if MODE == 'f16x3' else 8     # MFMAs per k-step of a hidden layer (x3: 4 row tiles x 3 products; f16 wide: 8 row tiles)
    simd = 32 * 16 * (2 * 6 * per_step + 5 * per_step)     # two waves per SIMD: 6 hidden layers each + 3 + 2 v-out units
    print(f'  MFMA issue cycles of one SIMD (two waves): {simd} ({100 * simd / tot.mean():.1f} % of the lifetime; ONE workgroup per CU)')
elif MODE == 'fp32':
    kl = [a[:, 10] - a[:, 2], a[:, 11] - a[:, 10], a[:, 12] - a[:, 11]]
    vl = [a[:, 13] - a[:, 5], a[:, 14] - a[:, 13], a[:, 15] - a[:, 14]]
    print('  k hidden layers 1..3: ' + ', '.join(f'{x.mean():.0f}' for x in kl) + '   v hidden layers 1..3: ' + ', '.join(f'{x.mean():.0f}' for x in vl))
    # by start order within the launch: the first 1024 workgroups start together (one round = 4 per CU x 256 CUs)
    order = np.argsort(a[:, 0])
    for nm, sel in (('first round (start in lockstep)', order[:1024]), ('later rounds', order[1024:])):
        dd = d[sel]
        print(f'  {nm}: lifetime {tot[sel].mean():.0f}; k hidden {dd[:, 2].mean():.0f}, v hidden {dd[:, 5].mean():.0f}, v out {dd[:, 6].mean():.0f}')
    mfma = 64 * (3 * 256 + 3 * 256 + 5 * 128)       # MFMA issue cycles of one wave (32-row workgroup): 6 hidden layers + its 5 v-out units
    print(f'  MFMA issue cycles of one wave: {mfma} ({100 * mfma / tot.mean():.1f} % of the lifetime; four workgroups share a CU)')
else:
    mfma = 32 * (6 * 128 + 5 * 64) * (1 if MODE == 'bf16-single' else 2)      # 128-row workgroup: 8 MFMAs per k-step and layer pass
    print(f'  MFMA issue cycles of one wave: {mfma} ({100 * mfma / tot.mean():.1f} % of the lifetime; two workgroups share a CU)')
# concurrency on a CU: how many workgroups ran on each CU and their span
key = (a[:, 9] & 0xF) << 32 | (a[:, 8] & 0xFF00)
for k in np.unique(key)[:3]:
    rows = a[key == k]
    rows = rows[np.argsort(rows[:, 0])]
    t0 = rows[0, 0]
    print(f'  CU {k:x}: {len(rows)} workgroups, span {rows[:, 7].max() - t0} ticks; starts {list((rows[:, 0] - t0)[:10])}')
