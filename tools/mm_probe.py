"""Developer probe: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the shapes of the two cs_attn contractions of a C3
tile in 16-bit -- a practical ceiling for a plain 16-bit MFMA GEMM on this part, next to which the hand-written kernels' fractions read.
   python tools/mm_probe.py"""
import torch, time
dev='cuda'
M,N,K=36864,1024,9216
A=torch.randn(M,K,device=dev,dtype=torch.float16)*0.01
B=torch.randn(N,K,device=dev,dtype=torch.float16)
for dt in (torch.float16, torch.bfloat16):
    a=A.to(dt); b=B.to(dt)
    for _ in range(3): c=a@b.t()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c=a@b.t()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(dt, f'{ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TFLOP/s')
# short-K scores GEMM shape: M=36864, N=9216, K=288
A=torch.randn(36864,288,device=dev,dtype=torch.float16); B=torch.randn(9216,288,device=dev,dtype=torch.float16)
for _ in range(3): c=A@B.t()
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): c=A@B.t()
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/20
print('scores shape', f'{ms:.3f} ms {2*36864*9216*288/ms/1e9:.0f} TFLOP/s')
