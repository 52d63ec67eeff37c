import sys, torch
sys.path.insert(0, ".")
from bench import rdn_ciaosr, time_steps
from ciaosr_amd import hip_ops
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
dev = torch.device("cuda:0")
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32)); seeded_init_(model, seed=0, gain=1.0); model = model.to(dev)
lq = synthetic_pair(1356, 2040, 4)[0].to(dev)
for prec in sys.argv[1:] or ["f16"]:
    model.restore(lq, options=prec)
    ms = time_steps(lambda: model.restore(lq, options=prec), 2, dev)
    with hip_ops.profile():
        model.restore(lq, options=prec); torch.cuda.synchronize()
    pr = hip_ops.profile.results()
    top = sorted(pr.items(), key=lambda kv: -kv[1]["total_ms"])[:4]
    print(f"C3 {prec}: {ms:8.1f} ms  " + " ".join(f"{k}={v['total_ms']:.0f}" for k, v in top), flush=True)
