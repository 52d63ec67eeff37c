#!/bin/bash
# K1 / K4 evidence of the staged routes on the C3 tile (one 192x192 LR tile, 589 824 queries in 20 eval_bsize chunks), on the GPU box:
#   bash tools/profile_staged.sh r6      (part of tools/profile_round.sh; ~3 min)
# per route: rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in their own --pmc passes (--kernel-trace only), the program
# directly after `--`.  Then tools/staged_summary.py turns gpurun_out/prof_<tag>/staged_* into the files committed under profiles/.
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
COMMON="--workload c3tile --steps 2 --warmup 1 --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe"
for ROUTE in as-written staged as-written-bf16 as-written-f16; do
  D=$O/staged_$ROUTE
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o s -- python3 $R/bench.py $COMMON --head-route $ROUTE > $D.bench.json 2> $D.stats.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -o f -- python3 $R/bench.py $COMMON --head-route $ROUTE > /dev/null 2> $D.fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -o w -- python3 $R/bench.py $COMMON --head-route $ROUTE > /dev/null 2> $D.write.err
  find $D -name "*kernel_trace.csv" -delete
  tail -2 $D.stats.err
done
python3 $R/tools/staged_summary.py $O $O/summary_staged
ls -la $O/summary_staged
