python tools/kernel_lab.py --quick fp32=fp32 2>&1 | tail -n 1 | cut -c1-400
CIAOSR_CSA_SOFTMAX_INPLACE=1 python tools/kernel_lab.py --quick fp32=fp32 2>&1 | tail -n 1 | cut -c1-400
