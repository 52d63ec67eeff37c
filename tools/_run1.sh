python bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline --no-live-pmc --no-extras 2>&1 | tail -n 1 | cut -c1-200
python tools/c5_breakdown.py 2>&1 | tail -n 6
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "swinir or e2e_restorer or encoder_features or narrow_trunk or csattn_small or gemm" 2>&1 | tail -n 3
