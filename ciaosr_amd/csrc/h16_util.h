// 16-bit MFMA operand types of the reduced-precision modes, as single gfx950 instructions.
//   F16 = false : bf16 (8 mantissa bits, fp32 range)          v_cvt_pk_bf16_f32      v_mfma_f32_32x32x16_bf16
//   F16 = true  : IEEE half (11 mantissa bits, |x| <= 65504)   v_cvt_pk_f16_f32       v_mfma_f32_32x32x16_f16
// Both MFMAs run at the same rate; conversions are round-to-nearest-even.  The half conversions SATURATE at +-65504
// (v_med3_f32 in front of the convert; a ReLU'd value gets relu and saturation from ONE v_med3_f32): an activation beyond
// the half range is clamped instead of becoming inf and then NaN in the next layer.
//   pack_h16x2   : two fp32 -> one packed pair (the software bf16 form costs ~5 VALU ops per element and made the bf16
//                  kernels VALU-bound: 11 VALU instructions per MFMA by SQ_INSTS_VALU)
//   quad_xor1/2  : DPP quad_perm moves instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32 (an LDS
//                  instruction per shuffle)
// The three 16-bit translation units (head_fused_h16.hip, gemm_h16.hip, dense_h16.hip) are compiled twice, with
// -DCIAOSR_F16=0 into namespace ciaosr::b16 and -DCIAOSR_F16=1 into ciaosr::f16 (Makefile); the small helpers of
// patch_ops.hip / head_ops.hip take the element type as a template argument.
#pragma once
#include <hip/hip_runtime.h>

#ifndef CIAOSR_F16
#define CIAOSR_F16 0
#endif
#if CIAOSR_F16
#define CIAOSR_H16_NS f16
#define CIAOSR_H16_SUFFIX "_f16"
#else
#define CIAOSR_H16_NS b16
#define CIAOSR_H16_SUFFIX "_bf16"
#endif

namespace ciaosr {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr float kHalfMax = 65504.f;

template <bool F16>
__device__ __forceinline__ unsigned pack_h16x2(float lo, float hi) {
    if constexpr (F16) {
        const f32x2_t v = {__builtin_amdgcn_fmed3f(lo, -kHalfMax, kHalfMax), __builtin_amdgcn_fmed3f(hi, -kHalfMax, kHalfMax)};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    } else {
        const f32x2_t v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    }
}
// relu(lo), relu(hi) packed
template <bool F16>
__device__ __forceinline__ unsigned pack_relu_h16x2(float lo, float hi) {
    if constexpr (F16) {
        const f32x2_t v = {__builtin_amdgcn_fmed3f(lo, 0.f, kHalfMax), __builtin_amdgcn_fmed3f(hi, 0.f, kHalfMax)};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    } else {
        const f32x2_t v = {fmaxf(lo, 0.f), fmaxf(hi, 0.f)};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    }
}
template <bool F16>
__device__ __forceinline__ uint2 pack_h16x4(float a, float b, float c, float d) {
    return make_uint2(pack_h16x2<F16>(a, b), pack_h16x2<F16>(c, d));
}
template <bool F16>
__device__ __forceinline__ unsigned short to_h16(float f) { return (unsigned short)(pack_h16x2<F16>(f, 0.f) & 0xFFFFu); }
// element 0 / element 1 of a packed pair back to fp32 (exact)
template <bool F16>
__device__ __forceinline__ float h16_lo(unsigned pair) {
    if constexpr (F16) return (float)__builtin_bit_cast(f16x2_t, pair).x;
    else return __uint_as_float(pair << 16);
}
template <bool F16>
__device__ __forceinline__ float h16_hi(unsigned pair) {
    if constexpr (F16) return (float)__builtin_bit_cast(f16x2_t, pair).y;
    else return __uint_as_float(pair & 0xFFFF0000u);
}
// D = A(32x16) . B(16x32) + C with 8 packed 16-bit elements per lane and operand
template <bool F16>
__device__ __forceinline__ f32x16_t mfma_h16(uint4 a, uint4 b, f32x16_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ float quad_xor1(float v) {     // value of lane ^ 1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {     // value of lane ^ 2
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}

}  // namespace ciaosr
