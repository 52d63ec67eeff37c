python -m pytest tests/test_hip_parity.py -q -x -k "csattn or full_c3_tile_vs_reference or precision_selects or head_bf16_mode" 2>&1 | tail -3
python3 tools/kernel_lab.py --quick f16=f16 bf16=bf16 2>&1 | tail -2
