#!/bin/bash
# Round profile capture on the GPU box: bench lines + rocprofv3 kernel stats + HBM-traffic PMC passes.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r1'
# Writes gpurun_out/prof_<tag>/...; copy the summaries to profiles/ afterwards (tools/trace_summary.py, pmc_summary.py).
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/c2_bench.json 2> $O/c2_bench.err
python3 $R/bench.py --workload c3tile --steps 5 --warmup 2 > $O/c3tile_bench.json 2>> $O/c2_bench.err
python3 $R/bench.py --precision bf16 --no-cpu-baseline > $O/c2_bf16_bench.json 2>> $O/c2_bench.err
python3 $R/bench.py --precision bf16 --workload c3tile --steps 5 --warmup 2 > $O/c3tile_bf16_bench.json 2>> $O/c2_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o c2 -- python3 $R/bench.py --no-cpu-baseline > $O/c2_bench_under_rocprof.json 2> $O/c2_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3tile -o c3tile -- python3 $R/bench.py --no-cpu-baseline --workload c3tile --steps 5 --warmup 2 > $O/c3tile_bench_under_rocprof.json 2> $O/c3tile_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o q -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $O/pmc_sq.err
# keep the merge-back small: traces of the PMC passes are only needed as counter CSVs
find $O -name "*kernel_trace.csv" -path "*pmc*" -delete
ls -la $O $O/*
