// Does v_mfma_f32_32x32x16_f16 keep half SUBNORMAL inputs, and does v_cvt_pk_f16_f32 produce them?  (Decides whether the lo halves of
// activation pairs -- |lo| <= |a| 2^-11, subnormal in half for |a| < 0.125 -- survive the matrix pipe as they are.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/f16_denorm.hip -o tools/ubench/f16_denorm && tools/ubench/f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float a_val, float b_val, float* out) {
    const f32x2 av = {a_val, a_val}, bv = {b_val, b_val};
    const f16x2 ah = __builtin_convertvector(av, f16x2), bh = __builtin_convertvector(bv, f16x2);
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = ah.x; b[i] = bh.x; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = c[0];                 // 16 products a*b
        out[1] = (float)ah.x;          // what the conversion made of a
        out[2] = (float)ah.x * (float)bh.x * 16.f;
    }
}

int main() {
    float* d;
    hipMalloc(&d, 64);
    const float as[] = {1.0f, 3.0517578125e-05f /*2^-15: subnormal*/, 9.5367431640625e-07f /*2^-20*/, 5.9604644775390625e-08f /*2^-24: smallest*/,
                        2.98e-08f /* below half the smallest: rounds to 0 */};
    for (float a : as) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, 1024.f, d);
        float h[3];
        hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a = %.6e  cvt -> %.6e   mfma(16 x a x 1024) = %.6e   expected %.6e   %s\n", a, h[1], h[0], h[2],
               h[0] == h[2] ? "kept" : "FLUSHED / differs");
    }
    return 0;
}
