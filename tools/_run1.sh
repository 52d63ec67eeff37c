python tools/enc_lab.py fp32 8 2>&1 | tail -n 1
for v in 1 4; do CIAOSR_HIP_LIB=$PWD/ciaosr_amd/csrc/libciaosr_st$v.so python tools/enc_lab.py fp32 8 2>&1 | tail -n 1; done
CIAOSR_WINO_PERSIST=0 python tools/enc_lab.py fp32 8 2>&1 | tail -n 1
