python tools/kernel_lab.py --quick fp32=fp32 2>&1 | tail -n 1 | cut -c1-260
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "csattn or full_c3_tile or e2e_restorer" 2>&1 | tail -n 3
