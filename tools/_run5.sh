#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for TB in 7 14; do
for P in f16 fp32; do
  timeout 900 python bench.py --workload c3 --precision $P --tile-batch $TB --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc --no-rccl-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('tile_batch $TB', '$P', d['ms_per_step'], {n:v for n,v in k.items() if n.startswith('enc_')})"
done
done
