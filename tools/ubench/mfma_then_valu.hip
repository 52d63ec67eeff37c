// Microbenchmark (round 4): does an fp32 VALU stream issued right behind a wave's own fp32 MFMAs compute the right values, and do the
// MFMAs?  (dense_wino4_f32.hip: a T phase that starts < ~128 cycles behind the M phase's last MFMA gives wrong images.)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_then_valu.hip -o tools/ubench/mfma_then_valu && tools/ubench/mfma_then_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF4 "v_mfma_f32_32x32x2_f32 %0, %14, %15, %0\nv_mfma_f32_32x32x2_f32 %1, %14, %15, %1\nv_mfma_f32_32x32x2_f32 %2, %14, %15, %2\nv_mfma_f32_32x32x2_f32 %3, %14, %15, %3\n"
// the 12-operation B^T column transform of F(4x4,3x3): x0..x5 = %16..%21 -> y0..y5 = %4..%9 (t1..t4 = %10..%13)
#define BT "v_fma_f32 %10, -4.0, %18, %20\nv_fma_f32 %11, -4.0, %17, %19\nv_sub_f32 %12, %20, %18\nv_sub_f32 %13, %19, %17\n" \
           "v_mov_b32 %4, %20\nv_fmac_f32 %4, 0xc0a00000, %18\nv_fmac_f32 %4, 4.0, %16\nv_add_f32 %5, %10, %11\nv_sub_f32 %6, %10, %11\n" \
           "v_fma_f32 %7, 2.0, %13, %12\nv_fma_f32 %8, -2.0, %13, %12\nv_mov_b32 %9, %21\nv_fmac_f32 %9, 0xc0a00000, %19\nv_fmac_f32 %9, 4.0, %17\n"
#define NOP6 "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"

template <int MODE> __global__ void run(float* out) {
    const int lane = threadIdx.x;
    const float a = 1.f + 0.01f * lane, b = 2.f - 0.02f * lane;
    float x[6], y[6] = {}, t[4] = {};
    for (int i = 0; i < 6; ++i) x[i] = 0.37f * (i + 1) + 0.011f * lane * (i + 2);
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
#define OPS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) \
            : "v"(a), "v"(b), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5])
    if (MODE == 0) asm volatile(MF4 BT NOP6 OPS);             // VALU right behind the MFMAs
    else           asm volatile(MF4 NOP6 BT "s_nop 15\n" OPS);  // the same VALU after the pipe has drained
    for (int i = 0; i < 6; ++i) out[MODE * 4096 + i * 64 + lane] = y[i];
    for (int i = 0; i < 16; ++i) out[MODE * 4096 + 1024 + i * 64 + lane] = c3[i] + c0[i];
}

int main() {
    float* d; hipMalloc(&d, 2 * 4096 * 4); hipMemset(d, 0, 2 * 4096 * 4);
    hipLaunchKernelGGL(run<0>, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(run<1>, dim3(1), dim3(64), 0, 0, d);
    static float h[2 * 4096];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int badv = 0, badm = 0;
    for (int i = 0; i < 6 * 64; ++i) badv += h[i] != h[4096 + i];
    for (int i = 0; i < 16 * 64; ++i) badm += h[1024 + i] != h[4096 + 1024 + i];
    printf("VALU results differing between 'right behind the MFMAs' and 'after the pipe drained': %d of 384 (y0 lane 1: %g vs %g)\n", badv, h[1], h[4097]);
    printf("MFMA results differing: %d of 1024\n", badm);
    return 0;
}
