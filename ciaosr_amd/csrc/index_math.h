// Exact fp32 coordinate / index arithmetic of the head.
// Every operation is an individually rounded fp32 op (__f*_rn: no FMA contraction), because the
// nearest-neighbour index is discontinuous and non-integer scales hit exact rounding ties.
#pragma once
#include <hip/hip_runtime.h>

namespace ciaosr {

// HIP's __fmul_rn/__fadd_rn are plain operators and may be contracted into FMAs under the default
// -ffp-contract=fast-honor-pragmas; these helpers carry the pragma that forbids it.
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float div_rn(float a, float b) {
#pragma clang fp contract(off)
    return a / b;
}

// F.grid_sample(mode='nearest', align_corners=False) source index on an axis of n samples:
// u = (c + 1) * (n / 2) - 0.5 (three roundings), idx = round-half-even(u).   (ciaosr_net.py:145)
__device__ __forceinline__ int nearest_index(float c, int n) {
    const float u = sub_rn(mul_rn(add_rn(c, 1.0f), (float)n * 0.5f), 0.5f);
    return (int)__builtin_rintf(u);
}

// shifted + clamped key coordinate along one axis (ciaosr_net.py:162-173):
//   t = (n - 1) / (1 - cell0);  r = 1 / t;  c' = clamp(c + (sign * r + 1e-6), -1 + 1e-6, 1 - 1e-6)
__device__ __forceinline__ float shifted_coord(float c, float cell0, int n, int sign) {
    float out = c;
    if (sign != 0) {
        const float t = div_rn((float)(n - 1), sub_rn(1.0f, cell0));
        const float r = div_rn(1.0f, t);
        const float d = add_rn(mul_rn((float)sign, r), 1e-6f);
        out = add_rn(c, d);
    }
    const float lo = (float)(-1 + 1e-6), hi = (float)(1 - 1e-6);
    return fminf(fmaxf(out, lo), hi);
}

// make_coord value of LR index k on an axis of n (mmedit make_coord, call site ciaosr_net.py:148)
__device__ __forceinline__ float pixel_centre(int k, int n) {
    const float v0r = (float)(-1.0 + 1.0 / (double)n);
    const float r2 = (float)(2.0 / (double)n);
    return add_rn(v0r, mul_rn(r2, (float)k));
}

// shift list of query_rgb (ciaosr_net.py:152-155): local_size 1 -> {0}; 2 -> {-1,+1}^2; 3 -> {-1,0,1}^2
__device__ __forceinline__ void shift_of(int j, int local_size, int& sy, int& sx) {
    if (local_size == 1) { sy = 0; sx = 0; return; }
    const int n = (local_size == 2) ? 2 : 3;
    const int step = 4 - local_size;  // 2 or 1
    sy = -1 + (j / n) * step;
    sx = -1 + (j % n) * step;
}

struct KeySample {
    int ky, kx;
    float rel_y, rel_x;
};

// key sample j of a query at (cy,cx): nearest LR pixel of the shifted coordinate and the relative
// offset fed to the MLPs: rel = (coord - coord_k) * [H, W]   (ciaosr_net.py:176-189)
__device__ __forceinline__ KeySample key_sample(float cy, float cx, float cell0y, float cell0x, int H, int W,
                                                int j, int local_size) {
    int sy, sx;
    shift_of(j, local_size, sy, sx);
    const float ky_c = shifted_coord(cy, cell0y, H, sy);
    const float kx_c = shifted_coord(cx, cell0x, W, sx);
    KeySample s;
    s.ky = nearest_index(ky_c, H);
    s.kx = nearest_index(kx_c, W);
    // grid_sample zero-pads out-of-range indices; after the clamp they cannot occur, but stay safe
    s.ky = min(max(s.ky, 0), H - 1);
    s.kx = min(max(s.kx, 0), W - 1);
    s.rel_y = mul_rn(sub_rn(cy, pixel_centre(s.ky, H)), (float)H);
    s.rel_x = mul_rn(sub_rn(cx, pixel_centre(s.kx, W)), (float)W);
    return s;
}

}  // namespace ciaosr
