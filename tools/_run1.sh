for a in 0 14 30; do echo "abl $a"; CIAOSR_GEMM_ABL=$a python tools/kernel_lab.py --quick fp32=fp32 2>&1 | tail -n 1 | grep -o "csa_attn_v=[0-9.]*"; done
