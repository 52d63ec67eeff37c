python tools/enc_lab.py fp32 8 2>&1 | tail -n 1
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "rdn_trunk or encoder_features or tile_batch" 2>&1 | tail -n 3
