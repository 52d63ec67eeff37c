python tools/enc_lab.py f16 8 2>&1 | tail -n 1
python tools/enc_lab.py bf16 8 2>&1 | tail -n 1
python tools/enc_lab.py f16 1 2>&1 | tail -n 1
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "rdn_trunk or tile_batch or full_c3_tile" 2>&1 | tail -n 3
