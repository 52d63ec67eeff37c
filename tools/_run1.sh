timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "whole_image_path_with" -s 2>&1 | grep -v "amdgpu.ids" | tail -n 4
