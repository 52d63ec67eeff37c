import torch, time, sys
sys.path.insert(0, ".")
from bench import rdn_ciaosr
from ciaosr_amd import hip_ops
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
dev=torch.device("cuda:0")
model=rdn_ciaosr(dict(scale=4,tile=192,tile_overlap=32)); seeded_init_(model,0); model=model.to(dev)
lq,_=synthetic_pair(48,48,4); lq=lq.to(dev)
ref=model.restore(lq).clone()
for mode in ("fp32","bf16"):
    model.test_cfg["precision"]=mode
    eager=model.restore(lq).clone()
    run=model.graphed_restore(lq)
    out=run()
    torch.cuda.synchronize()
    print(mode, "graph == eager bitwise:", torch.equal(out, eager))
    lq2=(lq*0.9).contiguous()
    o2=run(lq2).clone(); e2=model.restore(lq2)
    print(mode, "new input via static buffer:", torch.equal(o2, e2))
    for name, fn in (("eager", lambda: model.restore(lq)), ("graph", run)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(30): fn()
        torch.cuda.synchronize(); print(mode, name, round((time.perf_counter()-t)/30*1e3,3), "ms")
