python tools/kernel_lab.py --quick fp32=fp32 kv32=fp32,kv_rows=32 2>&1 | tail -n 2 | cut -c1-170
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "head or fused or full_c3_tile or e2e_restorer or determin or tile_batch" 2>&1 | tail -n 3
