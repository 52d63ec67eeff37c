#!/bin/bash
# SQ stall breakdown of the kernels of one precision at the C3 tile: bash tools/pmc_sq.sh fp32|bf16|f16 (own --pmc passes, --kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=${1:-bf16}
O=$R/gpurun_out/pmc_$P
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq1 -o s -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --precision $P --steps 2 --warmup 1 > /dev/null 2> $O/sq1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -o s -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --precision $P --steps 2 --warmup 1 > /dev/null 2> $O/sq2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --no-live-pmc --no-cpu-baseline --no-extras --no-rccl-probe --workload c3tile --precision $P --steps 2 --warmup 1 > /dev/null 2> $O/fetch.err
find $O -name "*kernel_trace.csv" -delete
tail -3 $O/sq1.err $O/sq2.err
