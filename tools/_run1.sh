python tools/kernel_lab.py --quick fp32=fp32 2>&1 | tail -n 1 | cut -c1-330
CIAOSR_HIP_LIB=$PWD/ciaosr_amd/csrc/libciaosr_bk16.so python tools/kernel_lab.py --quick fp32=fp32 2>&1 | tail -n 1 | cut -c1-330
CIAOSR_HIP_LIB=$PWD/ciaosr_amd/csrc/libciaosr_bk16.so timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gemm or csattn" 2>&1 | tail -n 3
