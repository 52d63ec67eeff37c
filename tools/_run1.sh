timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "logit_table_winograd" -s 2>&1 | tail -n 8
