#!/bin/bash
# round-6 GPU call 2: bf16-single rounding lab, fp32 SQ detail, C1 / C5 breakdowns (+ rocprof stats of each)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6b
mkdir -p $O
cd $R
python tools/bf16_single_lab.py > $O/bf16_single_lab.txt 2>&1
tail -8 $O/bf16_single_lab.txt
bash tools/pmc_sq.sh fp32 > $O/pmc_sq_fp32.log 2>&1
python tools/pmc_sq_detail.py gpurun_out/pmc_fp32 > $O/c3tile_fp32_sq_detail.txt 2>&1
head -30 $O/c3tile_fp32_sq_detail.txt
python tools/c1_breakdown.py > $O/c1_breakdown.txt 2>&1
python tools/c5_breakdown.py fp32 > $O/c5_breakdown.txt 2>&1
python tools/c5_breakdown.py f16 >> $O/c5_breakdown.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c1 -o c1 -- python3 $R/tools/c1_breakdown.py > /dev/null 2> $O/c1_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o c5 -- python3 $R/tools/c5_breakdown.py fp32 > /dev/null 2> $O/c5_rocprof.err
find $O -name "*kernel_trace.csv" -delete
tail -20 $O/c1_breakdown.txt $O/c5_breakdown.txt
