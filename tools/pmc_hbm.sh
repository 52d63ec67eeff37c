#!/bin/bash
# HBM traffic of the kernels of one precision at the C3 tile: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only).
#   bash tools/pmc_hbm.sh f16|bf16|f16-pairs|f16x3   ->  gpurun_out/pmc_hbm_<precision>/{fetch,write}; summarise with tools/pmc_summary.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=${1:-f16}
O=$R/gpurun_out/pmc_hbm_$P
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --precision $P --steps 2 --warmup 1 > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --precision $P --steps 2 --warmup 1 > /dev/null 2> $O/write.err
find $O -name "*kernel_trace.csv" -delete
tail -2 $O/fetch.err $O/write.err
