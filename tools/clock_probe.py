"""Developer probe: shader clock (rocm-smi) sampled while the fp32 head kernel runs back to back on one 192x192 tile.
   python tools/clock_probe.py"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import rdn_ciaosr
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair

dev = torch.device('cuda:0')
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32))
seeded_init_(model, seed=0, gain=1.0)
model = model.to(dev)
lq = synthetic_pair(192, 192, 4)[0].to(dev)
for _ in range(3):
    model.restore(lq)
torch.cuda.synchronize()
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=10).stdout
            samples.append([l.strip() for l in out.splitlines() if 'sclk' in l or 'Power' in l or 'fclk' in l or 'mclk' in l])
        except Exception as e:      # noqa: BLE001
            samples.append([repr(e)])
        time.sleep(0.2)


th = threading.Thread(target=sampler)
th.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(10):
        model.restore(lq)
    torch.cuda.synchronize()
    n += 10
el = time.perf_counter() - t0
stop = True
th.join()
print(f'{n} tiles in {el:.2f} s = {el / n * 1e3:.2f} ms/tile')
for s in samples[:3] + samples[-3:]:
    print(s)
