#!/bin/bash
# developer A/B of the 16-bit dense layer: C3 step time and the enc_dense share in f16 (and bf16 with "$1" = all)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for P in f16 ${1:+bf16}; do
  timeout 600 python bench.py --workload c3 --precision $P --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc --no-rccl-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('$P', d['ms_per_step'], {n:v for n,v in k.items() if n.startswith('enc_dense')})"
done
