"""Developer probe: per-wave phase stamps of dense_wino_f32_kernel on a batch of C3 tiles (needs `make -C ciaosr_amd/csrc probe`).
   CIAOSR_HIP_LIB=ciaosr_amd/csrc/libciaosr_hip_probe.so python tools/wino_probe.py [groups] [batch]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import rdn_ciaosr
from ciaosr_amd import _lib, hip_ops
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair

dev = torch.device('cuda')
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, 0)
model = model.to(dev)
enc = model.generator._encoder_hip
x = synthetic_pair(192, 192, 4)[0].expand(B, -1, -1, -1).contiguous().to(dev)
lib = _lib.load()
reader = lib.ciaosr_debug_wino_probe_read
reader.restype = C.c_int
N = 1024 * 8 * 32
buf = (C.c_ulonglong * N)()
assert reader(buf, 8, G) == 0          # select the launches to stamp
for _ in range(2):
    enc.forward_hwc_batch(x, hip_ops.Options('fp32'))
torch.cuda.synchronize()
assert reader(buf, N, G) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 32).astype(np.int64)
n_wg = 288
a = a[:n_wg]
TICK = 1.0      # the counter runs at the shader clock here
t0 = a[:, :, 0].min(axis=1, keepdims=True)
def rel(slot):
    return (a[:, :, slot] - t0) * TICK
life = (a[:, :, 30].max(axis=1) - a[:, :, 0].min(axis=1))
print(f'G = {G}, batch {B}: middle image, {n_wg} workgroups; lifetime avg {life.mean():.0f} cycles (min {life.min():.0f}, max {life.max():.0f})')
print(f'  prologue: issue done {rel(1).mean():.0f}, own DMA + weights landed {rel(2).mean():.0f} (max over waves {rel(2).max(axis=1).mean():.0f}), past the barrier {rel(3).mean():.0f}')
prev = rel(3)
for g in range(min(G, 2)):
    for jj in range(8):
        e = rel(4 + 8 * g + jj)
        dt = e - prev
        # waves w and w + 4 share a SIMD
        print(f'  group {g} step {jj}: {dt.mean():6.0f}  (waves 0-3 {dt[:, :4].mean():6.0f}, waves 4-7 {dt[:, 4:].mean():6.0f}; min {dt.min(axis=1).mean():6.0f} max {dt.max(axis=1).mean():6.0f})'
              f'   cumulative {np.mean(e - rel(3)):7.0f}')
        prev = e
    b = rel(20 + g)
    print(f'  group {g}: vmcnt + barrier wait {np.mean(b - prev):6.0f} (max over waves {np.mean((b - prev).max(axis=1)):6.0f}, min {np.mean((b - prev).min(axis=1)):6.0f})')
    prev = b
print(f'  output transform: regs -> LDS {np.mean(rel(28) - prev):.0f}, barrier {np.mean(rel(29) - rel(28)):.0f}, LDS -> global {np.mean(rel(30) - rel(29)):.0f}')
mf = 16 * 64
print(f'  MFMA issue cycles of one wave per step: {mf}; two waves share a SIMD -> {2 * mf} per step')
one = a[5]
print('  one workgroup, step-end stamps per wave (cycles from its first stamp):')
for wv in range(8):
    print(f'    wave {wv}: ' + ' '.join(f'{int(one[wv, 4 + k] - one[:, 0].min()):6d}' for k in range(8 * min(G, 2))))
