python tools/enc_lab.py fp32 8 2>&1 | tail -n 1
python tools/kernel_lab.py --quick fp32=fp32 2>&1 | tail -n 1 | cut -c1-260
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "rdn_trunk or encoder_features or fused_and_staged or full_c3_tile or tile_batch or tile_streams or e2e_restorer" 2>&1 | tail -n 3
