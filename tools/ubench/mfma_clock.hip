// Microbenchmark (round 4): the shader clock an fp32-MFMA-bound kernel really runs at.  One wave per SIMD issues N independent
// v_mfma_f32_32x32x2_f32 back to back (64 cycles each); time / (N * 64) = the clock.  Run on 1, 64 and 256 CUs and for 16-bit MFMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_clock.hip -o tools/ubench/mfma_clock && tools/ubench/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    const float x = 1.f + threadIdx.x * 1e-3f, y = 2.f - threadIdx.x * 1e-3f;
    f16x8 hx, hy;
    for (int i = 0; i < 8; ++i) { hx[i] = (_Float16)(x + i); hy[i] = (_Float16)(y - i); }
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        } else if (KIND == 2) {       // dependent chain: every MFMA accumulates into the same tile
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a0, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, a3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND> void run(const char* name, int wgs, int iters, int cyc) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, d, iters);       // warm
    hipDeviceSynchronize();
    float best = 1e30f, last = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&last, e0, e1); if (last < best) best = last;
    }
    const double cycles = (double)iters * 4 * cyc;
    printf("%-10s %3d workgroups (one per CU): %8.3f ms best, %8.3f ms last of 5 -> %.3f GHz (%.3f GHz in the last, sustained run)\n", name, wgs, best, last,
           cycles / (best * 1e-3) * 1e-9, cycles / (last * 1e-3) * 1e-9);
    hipFree(d);
}

int main() {
    for (int wgs : {1, 64, 256}) run<0>("fp32 MFMA", wgs, 200000, 64);
    for (int wgs : {1, 256}) run<1>("f16 MFMA", wgs, 400000, 32);
    run<2>("fp32 dep.", 1, 200000, 64);           // four MFMAs in a row on ONE accumulator tile: does the dependent issue cost anything?
    // sustained: 2 s of fp32 MFMA on every CU
    run<0>("fp32 2s", 256, 4000000, 64);
    return 0;
}
