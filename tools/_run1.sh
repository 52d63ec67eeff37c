timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "past_4gib or gemm" -s 2>&1 | grep -v "amdgpu.ids" | tail -n 8
