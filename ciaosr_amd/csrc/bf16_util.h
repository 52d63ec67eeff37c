// bf16 conversion and quad shuffles as single gfx950 instructions.
//   pack_bf16x2  : v_cvt_pk_bf16_f32 (round-to-nearest-even, same result as the integer trick for finite inputs;
//                  the software form costs ~5 VALU ops per element and made the bf16 kernels VALU-bound: 11 VALU
//                  instructions per MFMA by SQ_INSTS_VALU)
//   quad_xor1/2  : DPP quad_perm moves instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32 (an LDS
//                  instruction per shuffle)
#pragma once
#include <hip/hip_runtime.h>

namespace ciaosr {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) {
    return make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
__device__ __forceinline__ unsigned short to_bf16(float f) { return (unsigned short)(pack_bf16x2(f, 0.f) & 0xFFFFu); }

__device__ __forceinline__ float quad_xor1(float v) {     // value of lane ^ 1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {     // value of lane ^ 2
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}

}  // namespace ciaosr
