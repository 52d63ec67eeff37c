timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gemm or csattn or head or full_c3_tile or e2e_restorer or swinir" 2>&1 | tail -n 3
python bench.py --no-extras --no-cpu-baseline --no-live-pmc 2>&1 | tail -n 1 | cut -c1-330
