"""Developer probe: what clock and power the GPU runs at under one kernel mix.  Loops `restore()` on a 192x192 tile in the given precision
for SECONDS while sampling `rocm-smi` (sclk, power, temperature) once a second from a child process.
   python tools/clock_probe.py [f16|bf16|fp32] [seconds]"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rdn_ciaosr                              # noqa: E402
from ciaosr_amd import hip_ops                            # noqa: E402
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair   # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'f16'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
dev = torch.device('cuda')
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, 0)
model = model.to(dev)
lq = synthetic_pair(192, 192, 4)[0].to(dev)
opt = hip_ops.Options(mode)
for _ in range(3):
    model.restore(lq, options=opt)
torch.cuda.synchronize()
print(subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--showmaxpower'], capture_output=True, text=True).stdout[-1500:], flush=True)
t0 = time.time()
n = 0
samples = []
while time.time() - t0 < secs:
    for _ in range(20):
        model.restore(lq, options=opt)
    n += 20
    if len(samples) < int(time.time() - t0):          # queue stays full while rocm-smi runs
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
        samples.append(' | '.join(ln.strip() for ln in out.splitlines() if 'sclk' in ln or 'Power' in ln or 'mclk' in ln))
torch.cuda.synchronize()
dt = time.time() - t0
print(f'{mode}: {n} tiles in {dt:.2f} s = {1e3 * dt / n:.3f} ms per tile')
for s in samples:
    print(s)
