python -m pytest tests/test_hip_parity.py -q -x -s -k "full_c3_tile_vs_reference and f16-pairs" 2>&1 | grep "tile192\|passed\|failed\|Error" | cut -c1-420
