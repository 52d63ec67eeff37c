for v in A B; do
echo "variant $v"
CIAOSR_HIP_LIB=$PWD/ciaosr_amd/csrc/variants/libslp$v.so python -m pytest tests/test_hip_parity.py -q -k "head_rerun_is_bitwise_deterministic_at_scale" 2>&1 | tail -4
CIAOSR_HIP_LIB=$PWD/ciaosr_amd/csrc/variants/libslp$v.so python -m pytest tests/test_hip_parity.py -q -k "head_rerun_is_bitwise_deterministic_at_scale" 2>&1 | tail -2
done
