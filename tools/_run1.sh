python -m pytest tests/test_hip_parity.py -q -x -s -k "csattn" 2>&1 | grep "composed tail\|passed\|failed\|Error\|assert" | head -20
python3 tools/kernel_lab.py --quick fp32=fp32 gemm=fp32,csa_scores_gemm=1 2>&1 | tail -2
