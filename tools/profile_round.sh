#!/bin/bash
# Round profile capture on the GPU box: default bench line (C3) + rocprofv3 kernel stats + HBM-traffic PMC passes.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r2 [quick]'
# Writes gpurun_out/prof_<tag>/...; copy the summaries to profiles/ afterwards (tools/trace_summary.py, pmc_summary.py).
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/c3_bench.json 2> $O/c3_bench.err
python3 $R/bench.py --workload c3tile --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-rccl-probe > $O/c3tile_bench.json 2>> $O/c3_bench.err
python3 $R/bench.py --workload c3tile --precision bf16 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-rccl-probe > $O/c3tile_bf16_bench.json 2>> $O/c3_bench.err
python3 $R/bench.py --workload c3tile --precision f16 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-rccl-probe > $O/c3tile_f16_bench.json 2>> $O/c3_bench.err
python3 $R/bench.py --workload c3 --precision f16 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc --no-rccl-probe > $O/c3_f16_bench.json 2>> $O/c3_bench.err
python3 $R/bench.py --workload c3 --precision f16-pairs --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc --no-rccl-probe > $O/c3_f16_pairs_bench.json 2>> $O/c3_bench.err
python3 $R/bench.py --workload c3 --precision bf16 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc --no-rccl-probe > $O/c3_bf16_bench.json 2>> $O/c3_bench.err
python3 $R/bench.py --workload c3 --precision bf16 --bf16-single --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc --no-rccl-probe > $O/c3_bf16_single_bench.json 2>> $O/c3_bench.err
python3 $R/bench.py --workload c3 --precision f16x3 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc --no-rccl-probe > $O/c3_f16x3_bench.json 2>> $O/c3_bench.err
python3 $R/bench.py --workload c3 --precision f16x3-fast --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc --no-rccl-probe > $O/c3_f16x3_fast_bench.json 2>> $O/c3_bench.err
python3 $R/bench.py --workload c3tile --precision f16x3 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-rccl-probe > $O/c3tile_f16x3_bench.json 2>> $O/c3_bench.err
python3 $R/bench.py --workload c2 --steps 20 --warmup 5 > $O/c2_bench.json 2>> $O/c3_bench.err
# kernel stats of the SAME command as the driver's bench, on the 6-tile variant of C3 and on one tile (trace size bounded)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3tile -o c3tile -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --steps 5 --warmup 2 > $O/c3tile_bench_under_rocprof.json 2> $O/c3tile_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3tile_bf16 -o c3tile_bf16 -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --precision bf16 --steps 5 --warmup 2 > $O/c3tile_bf16_bench_under_rocprof.json 2> $O/c3tile_bf16_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3tile_f16 -o c3tile_f16 -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --precision f16 --steps 5 --warmup 2 > $O/c3tile_f16_bench_under_rocprof.json 2> $O/c3tile_f16_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3tile_f16x3 -o c3tile_f16x3 -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --precision f16x3 --steps 5 --warmup 2 > $O/c3tile_f16x3_bench_under_rocprof.json 2> $O/c3tile_f16x3_rocprof.err
if [ "$2" != "quick" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o c2 -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c2 --steps 20 --warmup 5 > $O/c2_bench_under_rocprof.json 2> $O/c2_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --steps 2 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --steps 2 --warmup 1 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o q -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --steps 2 --warmup 1 > /dev/null 2> $O/pmc_sq.err
fi
# K1 / K4 of the staged routes (north star: >= 40 % of the HBM roofline on the local-attention kernel) on the C3 tile: kernel stats + FETCH / WRITE
# counters of `--head-route as-written` (gather_rows_kernel = K1, local_attention_kernel<4> = K4), `staged` (head_rows_kernel = the hoisted K1)
# and `as-written-bf16` / `-f16` (local_attention_h16_kernel); summarise with tools/staged_summary.sh -> profiles/<tag>_c3tile_staged_*
bash $R/tools/profile_staged.sh $TAG
# keep the merge-back small: the per-launch traces are only needed as the stats CSVs / counter CSVs
find $O -name "*kernel_trace.csv" -path "*pmc*" -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -la $O $O/*
