#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -s -k "wide_tiles or (full_c3_tile and (bf16 or f16)) or (trained_like_16bit and bf16-single)" > $O/t.log 2>&1
tail -15 $O/t.log
for P in f16 bf16 f16-pairs; do
  timeout 600 python bench.py --workload c3 --precision $P --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc --no-rccl-probe > $O/c3_$P.json 2> $O/c3_$P.err
  python - <<PY
import json
d=json.loads(open('$O/c3_$P.json').read().strip().splitlines()[-1])
k=d['kernels_ms_per_step']
print('$P', d['ms_per_step'], {n:v for n,v in list(k.items())[:6]})
PY
done
timeout 600 python bench.py --workload c3 --precision bf16 --bf16-single --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc --no-rccl-probe > $O/c3_bf16_single.json 2> $O/c3_bf16_single.err
python -c "
import json
d=json.loads(open('$O/c3_bf16_single.json').read().strip().splitlines()[-1]); print('bf16-single', d['ms_per_step'], list(d['kernels_ms_per_step'].items())[:6])"
