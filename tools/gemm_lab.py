"""Developer lab: exact-fp32 GEMM (ciaosr_gemm_f32) at the attn.V shape for several row counts: workgroups per slot and TFLOP/s.
   python tools/gemm_lab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ciaosr_amd import hip_ops

dev = torch.device('cuda:0')
N, K = 1024, 9216
b = torch.randn(K, N, device=dev)
for tiles_m in (256, 272, 288, 304, 320, 512, 576):
    M = tiles_m * 128
    a = torch.randn(M, K, device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(2):
        hip_ops.gemm(a, b, b_is_kn=True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        hip_ops.gemm(a, b, b_is_kn=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    wgs = tiles_m * (N // 128)
    print(f'M = {M:6d}: {wgs:5d} workgroups = {wgs / 512:5.2f} rounds of 512 slots; {ms:7.3f} ms = {2.0 * M * N * K / ms / 1e9:6.1f} TFLOP/s '
          f'({2.0 * M * N * K / ms / 1e9 / 157.3:.3f} of the fp32 MFMA peak)', flush=True)
    del a, out
