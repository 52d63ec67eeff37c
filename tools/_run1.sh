python -m pytest tests/test_hip_parity.py -q -x -s -k "csattn or (full_c3_tile_vs_reference and fp32)" 2>&1 | grep "composed tail\|passed\|failed\|Error\|assert" | head -20
python3 tools/kernel_lab.py --quick fp32=fp32 2>&1 | tail -1
