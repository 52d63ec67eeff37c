#!/bin/bash
# gpurun_out/prof_r6{,x} (tools/profile_round.sh r6, tools/profile_r6_extra.sh) -> the summaries committed under profiles/r6_*
cd "$(dirname "$0")/.."
O=gpurun_out/prof_r6; X=gpurun_out/prof_r6x
for m in "" _bf16 _f16 _f16x3; do
  cp $O/c3tile$m/c3tile${m}_kernel_stats.csv profiles/r6_c3tile${m}_kernel_stats.csv
  cp $O/c3tile${m}_bench.json profiles/r6_c3tile${m}_bench.json
done
cp $O/c3tile_bench_under_rocprof.json profiles/r6_c3tile_bench_under_rocprof.json
for m in "" _bf16 _bf16_single _f16 _f16_pairs _f16x3 _f16x3_fast; do cp $O/c3${m}_bench.json profiles/r6_c3${m}_bench.json; done
cp $O/c2_bench.json profiles/r6_c2_bench.json
cp $O/c2/c2_kernel_stats.csv profiles/r6_c2_kernel_stats.csv
python tools/pmc_summary.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv > profiles/r6_c3tile_pmc_hbm_traffic.json
python tools/pmc_sq_summary.py $O/pmc_sq/q_counter_collection.csv > profiles/r6_c3tile_pmc_mfma_busy.json
for f in c3tile_staged_kernel_stats.csv c3tile_staged_pmc_hbm_traffic.json c3tile_staged_rooflines.txt; do cp $O/summary_staged/$f profiles/r6_$f; done
cp $O/staged_as-written.bench.json profiles/r6_c3tile_staged_bench_under_rocprof.json
cp $X/c3tile_f16_pmc_hbm_traffic.json profiles/r6_c3tile_f16_pmc_hbm_traffic.json
cp $X/c3tile_bf16_pmc_hbm_traffic.json profiles/r6_c3tile_bf16_pmc_hbm_traffic.json
cp $X/c3tile_f16_sq_detail.txt profiles/r6_c3tile_f16_sq_detail.txt
cp $X/spill_report.txt profiles/r6_spill_report.txt
