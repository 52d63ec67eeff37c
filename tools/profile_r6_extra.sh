#!/bin/bash
# round-6 extras on the GPU box: HBM-traffic counters of the 16-bit modes at the C3 tile, SQ detail in f16, spill report
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r6x
mkdir -p $O
cd $R
for P in f16 bf16; do
  bash tools/pmc_hbm.sh $P > $O/pmc_hbm_$P.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_hbm_$P/fetch/f_counter_collection.csv gpurun_out/pmc_hbm_$P/write/w_counter_collection.csv > $O/c3tile_${P}_pmc_hbm_traffic.json
done
bash tools/pmc_sq.sh f16 > $O/pmc_sq_f16.log 2>&1
python tools/pmc_sq_detail.py gpurun_out/pmc_f16 > $O/c3tile_f16_sq_detail.txt 2>&1
python tools/spill_report.py > $O/spill_report.txt 2>&1
head -40 $O/c3tile_f16_sq_detail.txt
