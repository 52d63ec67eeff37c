#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6d
mkdir -p $O
cd $R
python tools/bf16_single_lab.py ef+bc:cal48:rne 2>&1 | grep -v amdgpu.ids | tee $O/lab.txt
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -s -k "wide_tiles or (full_c3_tile and (bf16 or f16))" > $O/t.log 2>&1
grep -E "^tile192|passed|failed|^dense" $O/t.log | cut -c1-330
for P in f16 bf16; do
  timeout 600 python bench.py --workload c3 --precision $P --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc --no-rccl-probe > $O/c3_$P.json 2> $O/c3_$P.err
  python - <<PY
import json
d=json.loads(open('$O/c3_$P.json').read().strip().splitlines()[-1])
k=d['kernels_ms_per_step']
print('$P', d['ms_per_step'], {n:v for n,v in list(k.items())[:4]})
PY
done
