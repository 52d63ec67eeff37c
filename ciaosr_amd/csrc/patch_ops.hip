// HBM-bound data-movement kernels: layout changes, implicit unfold / patch extraction,
// row softmax, the gather-form fold of conv_transpose2d, restorer plumbing.
// All accesses are float4 along the channel (innermost) dimension: one 3x3 / 6x6 tap of a
// channels-last map is C contiguous floats.
#include "h16_util.h"
#include "common.h"
#include "index_math.h"

namespace ciaosr {

// ---------------------------------------------------------------------------------------------
// [C][H][W] <-> [H][W][ld]   (32x32 LDS transpose tiles: coalesced on both sides)
// ---------------------------------------------------------------------------------------------
__global__ void nchw_to_hwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW, int ld) {
    __shared__ float tile[32][33];
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? src[(size_t)c * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < C) dst[(size_t)p * ld + c] = tile[tx][r];
    }
}

__global__ void hwc_to_nchw_kernel(const float* __restrict__ src, int ld, float* __restrict__ dst, int C, int HW) {
    __shared__ float tile[32][33];
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? src[(size_t)p * ld + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) dst[(size_t)c * HW + p] = tile[tx][r];
    }
}

// ---------------------------------------------------------------------------------------------
// reflect pad right/bottom up to the next multiple of the scale (pad < size)  (arch_csnln.py:438-444)
// ---------------------------------------------------------------------------------------------
__global__ void pad_reflect_kernel(const float* __restrict__ src, int ld_src, int H, int W, int C,
                                   float* __restrict__ dst, int Hp, int Wp) {
    const int c4n = C >> 2;
    const long n = (long)Hp * Wp * c4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const long pix = i / c4n;
        int x = (int)(pix % Wp), y = (int)(pix / Wp);
        if (y >= H) y = 2 * (H - 1) - y;
        if (x >= W) x = 2 * (W - 1) - x;
        reinterpret_cast<float4*>(dst)[i] =
            *reinterpret_cast<const float4*>(src + ((size_t)y * W + x) * ld_src + 4 * c4);
    }
}

// F.interpolate(scale_factor=1/s, 'bilinear', align_corners=False) (arch_csnln.py:474) for integer s: the source coordinate of
// output d is s (d + 0.5) - 0.5, i.e. the single pixel s d + (s-1)/2 for odd s and the midpoint of pixels s d + s/2 - 1, s d + s/2
// for even s: a 1x1 (s = 3) or 2x2 (s = 2, 4) mean.
__global__ void downsample_kernel(const float* __restrict__ src, int Hp, int Wp, int C, int s, float* __restrict__ dst) {
    const int c4n = C >> 2;
    const int Ho = Hp / s, Wo = Wp / s;
    const int o0 = (s - 1) >> 1, o1 = s >> 1;
    const long n = (long)Ho * Wo * c4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const long pix = i / c4n;
        const int x = (int)(pix % Wo), y = (int)(pix / Wo);
        const float4* sp = reinterpret_cast<const float4*>(src);
        const float4 a = sp[((size_t)(s * y + o0) * Wp + s * x + o0) * c4n + c4];
        const float4 b = sp[((size_t)(s * y + o0) * Wp + s * x + o1) * c4n + c4];
        const float4 c = sp[((size_t)(s * y + o1) * Wp + s * x + o0) * c4n + c4];
        const float4 d = sp[((size_t)(s * y + o1) * Wp + s * x + o1) * c4n + c4];
        float4 o;
        o.x = ((a.x + b.x) + (c.x + d.x)) * 0.25f;
        o.y = ((a.y + b.y) + (c.y + d.y)) * 0.25f;
        o.z = ((a.z + b.z) + (c.z + d.z)) * 0.25f;
        o.w = ((a.w + b.w) + (c.w + d.w)) * 0.25f;
        reinterpret_cast<float4*>(dst)[i] = o;
    }
}

// 2x2 mean == F.interpolate(scale_factor=0.5, 'bilinear', align_corners=False) (arch_csnln.py:474)
__global__ void avgpool2_kernel(const float* __restrict__ src, int Hp, int Wp, int C, float* __restrict__ dst) {
    const int c4n = C >> 2;
    const int Ho = Hp >> 1, Wo = Wp >> 1;
    const long n = (long)Ho * Wo * c4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const long pix = i / c4n;
        const int x = (int)(pix % Wo), y = (int)(pix / Wo);
        const float4* s = reinterpret_cast<const float4*>(src);
        const float4 a = s[((size_t)(2 * y) * Wp + 2 * x) * c4n + c4];
        const float4 b = s[((size_t)(2 * y) * Wp + 2 * x + 1) * c4n + c4];
        const float4 c = s[((size_t)(2 * y + 1) * Wp + 2 * x) * c4n + c4];
        const float4 d = s[((size_t)(2 * y + 1) * Wp + 2 * x + 1) * c4n + c4];
        float4 o;
        o.x = ((a.x + b.x) + (c.x + d.x)) * 0.25f;
        o.y = ((a.y + b.y) + (c.y + d.y)) * 0.25f;
        o.z = ((a.z + b.z) + (c.z + d.z)) * 0.25f;
        o.w = ((a.w + b.w) + (c.w + d.w)) * 0.25f;
        reinterpret_cast<float4*>(dst)[i] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// patch rows: one wavefront per output row
// ---------------------------------------------------------------------------------------------
struct PatchP {
    const float* src;
    float* out;
    int ld_src, Hs, Ws, Cs, k, stride, pad, OH, OW, ld_out;
    int normalize;
    float floor_;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void patch_rows_kernel(PatchP p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)p.OH * p.OW) return;
    const int oy = (int)(row / p.OW), ox = (int)(row % p.OW);
    const int y0 = oy * p.stride - p.pad, x0 = ox * p.stride - p.pad;
    const int c4n = p.Cs >> 2;
    const int n4 = p.k * p.k * c4n;
    float* o = p.out + (size_t)row * p.ld_out;

    auto fetch = [&](int t) -> float4 {
        const int tap = t / c4n, c4 = t - tap * c4n;
        const int i = tap / p.k, j = tap - i * p.k;
        const int y = y0 + i, x = x0 + j;
        if (y < 0 || y >= p.Hs || x < 0 || x >= p.Ws) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4*>(p.src + ((size_t)y * p.Ws + x) * p.ld_src + 4 * c4);
    };

    if (!p.normalize) {
        for (int t = lane; t < n4; t += 64) reinterpret_cast<float4*>(o)[t] = fetch(t);
        return;
    }
    float ss = 0.f;
    for (int t = lane; t < n4; t += 64) {
        const float4 v = fetch(t);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = wave_sum(ss);
    const float den = fmaxf(sqrtf(ss), p.floor_);   // max(||w||, escape_NaN)  (arch_csnln.py:494-496)
    for (int t = lane; t < n4; t += 64) {
        float4 v = fetch(t);
        v.x /= den; v.y /= den; v.z /= den; v.w /= den;
        reinterpret_cast<float4*>(o)[t] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// in-place row softmax over the first L of ld columns; pad columns are zeroed
// (F.softmax(yi*10, dim=1), arch_csnln.py:505; the x10 is applied by the producing GEMM)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, long rows, int L, int ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* s = S + (size_t)row * ld;
    const int n4 = ld >> 2;
    float m = -INFINITY;
    for (int t = lane; t < n4; t += 64) {
        const float4 v = reinterpret_cast<const float4*>(s)[t];
        const int c = 4 * t;
        if (c < L) m = fmaxf(m, v.x);
        if (c + 1 < L) m = fmaxf(m, v.y);
        if (c + 2 < L) m = fmaxf(m, v.z);
        if (c + 3 < L) m = fmaxf(m, v.w);
    }
    m = wave_max(m);
    float sum = 0.f;
    for (int t = lane; t < n4; t += 64) {
        const float4 v = reinterpret_cast<const float4*>(s)[t];
        const int c = 4 * t;
        if (c < L) sum += expf(v.x - m);
        if (c + 1 < L) sum += expf(v.y - m);
        if (c + 2 < L) sum += expf(v.z - m);
        if (c + 3 < L) sum += expf(v.w - m);
    }
    sum = wave_sum(sum);
    for (int t = lane; t < n4; t += 64) {
        float4 v = reinterpret_cast<const float4*>(s)[t];
        const int c = 4 * t;
        v.x = c < L ? expf(v.x - m) / sum : 0.f;
        v.y = c + 1 < L ? expf(v.y - m) / sum : 0.f;
        v.z = c + 2 < L ? expf(v.z - m) / sum : 0.f;
        v.w = c + 3 < L ? expf(v.w - m) / sum : 0.f;
        reinterpret_cast<float4*>(s)[t] = v;
    }
}

// Single-pass variant for rows of up to 64 x SMV float4: the whole row lives in registers (one HBM read, one write).
// OUT: 0 = fp32 in place; 1 / 2 = write bf16 / half probabilities to P (row stride ldp) instead.
constexpr int SMV = 36;          // 36 float4 per lane = rows of up to 9216 floats (the 192x192 tile's L)
template <int OUT>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(float* __restrict__ S, long rows, int L, int ld,
                                                               unsigned short* __restrict__ P, int ldp) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4* s = reinterpret_cast<float4*>(S + (size_t)row * ld);
    constexpr bool OUT16 = OUT != 0;
    const int n4 = (OUT16 ? ldp : ld) >> 2;
    float4 v[SMV];
#pragma unroll
    for (int i = 0; i < SMV; ++i) {
        const int t = lane + 64 * i, c = 4 * t;
        v[i] = (t < n4 && c < L) ? s[t] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (c + 1 >= L) v[i].y = -INFINITY;
        if (c + 2 >= L) v[i].z = -INFINITY;
        if (c + 3 >= L) v[i].w = -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < SMV; ++i) m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SMV; ++i) {                      // exp(-inf) = 0 for the masked tail
        v[i].x = expf(v[i].x - m); v[i].y = expf(v[i].y - m); v[i].z = expf(v[i].z - m); v[i].w = expf(v[i].w - m);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int i = 0; i < SMV; ++i) {
        const int t = lane + 64 * i;
        if (t >= n4) continue;
        const float4 o = make_float4(v[i].x / sum, v[i].y / sum, v[i].z / sum, v[i].w / sum);
        if (OUT16) {
            reinterpret_cast<uint2*>(P + (size_t)row * ldp)[t] = pack_h16x4<OUT == 2>(o.x, o.y, o.z, o.w);
        } else {
            s[t] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fold: gather form of F.conv_transpose2d(P, V, stride=2, padding=2) (arch_csnln.py:511).
// O[(y,x)][(i*6+j)*C + c] holds the 6x6xC patch weighted for LR pixel (y,x); output pixel (u,v) of
// the 2Hp x 2Wp map sums the <= 9 (pixel, tap) pairs with 2y-2+i == u, 2x-2+j == v.
// Deterministic (no atomics).
// ---------------------------------------------------------------------------------------------
__global__ void fold_kernel(const float* __restrict__ O, int ldo, int Hp, int Wp, int C, float* __restrict__ Y) {
    const int c4n = C >> 2;
    const int H2 = 2 * Hp, W2 = 2 * Wp;
    const long n = (long)H2 * W2 * c4n;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const long pix = idx / c4n;
        const int v = (int)(pix % W2), u = (int)(pix / W2);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = u & 1; i < 6; i += 2) {
            const int y = (u + 2 - i) >> 1;
            if (y < 0 || y >= Hp) continue;
            for (int j = v & 1; j < 6; j += 2) {
                const int x = (v + 2 - j) >> 1;
                if (x < 0 || x >= Wp) continue;
                const float4 t = *reinterpret_cast<const float4*>(
                    O + ((size_t)y * Wp + x) * ldo + (size_t)(i * 6 + j) * C + 4 * c4);
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
        }
        reinterpret_cast<float4*>(Y)[idx] = acc;
    }
}

// fold for a general scale s: F.conv_transpose2d(P, V, stride=s, padding=s) with (3s)x(3s) patches.
// O[(y,x)][(i*3s+j)*C + c]; output pixel (u,v) of the sHp x sWp map sums the <= 9 (pixel, tap) pairs with s y - s + i == u.
__global__ void fold_s_kernel(const float* __restrict__ O, int ldo, int Hp, int Wp, int C, int s, float* __restrict__ Y) {
    const int c4n = C >> 2;
    const int Hs = s * Hp, Ws = s * Wp, k = 3 * s;
    const long n = (long)Hs * Ws * c4n;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const long pix = idx / c4n;
        const int v = (int)(pix % Ws), u = (int)(pix / Ws);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = u % s; i < k; i += s) {
            const int y = (u + s - i) / s;            // u + s - i is a non-negative multiple of s or negative
            if (u + s - i < 0 || y >= Hp) continue;
            for (int j = v % s; j < k; j += s) {
                const int x = (v + s - j) / s;
                if (v + s - j < 0 || x >= Wp) continue;
                const float4 t = *reinterpret_cast<const float4*>(O + ((size_t)y * Wp + x) * ldo + (size_t)(i * k + j) * C + 4 * c4);
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
        }
        reinterpret_cast<float4*>(Y)[idx] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Composed fold + down-convolution of CrossScaleAttention (exact).
// conv_transpose2d(stride 2, pad 2) followed by the 3x3 stride-2 `down` conv means output pixel (y',x') only
// receives from the 6x6 patches of LR pixels (y'+dy, x'+dx), dy,dx in {-2..1}, and the patch taps involved
// collapse, through `down`'s weights, to C numbers per (l, offset):
//   V'[l][(dy,dx)][co] = Pc_{R(dy),S(dx)}[(ly-dy, lx-dx)][co],
//   Pc_{R,S}[l'][co]   = sum_ci sum_{a in R, b in S} Wd[co,ci,a,b] * E0[ci, 2ly'-1+a, 2lx'-1+b]   (E0 = E, zero outside)
//   R(-2) = {0}, R(-1) = R(0) = {0,1,2}, R(1) = {1,2}
// so attn.V shrinks from N = 36C to N = 16C columns.  Output row 0 / column 0 must not see the cropped row -1 of
// the 2x map: there the dy = 0 (dx = 0) blocks use the {1,2} subset instead ("top", "left", "corner" variants).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int b3_subset(int d) { return d == -2 ? 0 : (d == 1 ? 2 : 1); }   // {0} / {0,1,2} / {1,2}

// V'[l][blk*C + co]: blk 0..15 main (dy+2)*4+(dx+2); 16..19 top variants (0,dx); 20..23 left variants (dy,0); 24 corner
__global__ void csa_gather_vprime_kernel(const float* __restrict__ Pc, int Hh, int Wh, int C, float* __restrict__ Vp) {
    const int c4n = C >> 2;
    const long n = (long)Hh * Wh * 25 * c4n;
    const int We = Wh + 3;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const long t = idx / c4n;
        const int blk = (int)(t % 25);
        const int l = (int)(t / 25);
        const int ly = l / Wh, lx = l - ly * Wh;
        int dy, dx, r, sct;
        if (blk < 16) { dy = blk / 4 - 2; dx = blk % 4 - 2; r = b3_subset(dy); sct = b3_subset(dx); }
        else if (blk < 20) { dy = 0; dx = blk - 16 - 2; r = 2; sct = b3_subset(dx); }
        else if (blk < 24) { dy = blk - 20 - 2; dx = 0; r = b3_subset(dy); sct = 2; }
        else { dy = 0; dx = 0; r = 2; sct = 2; }
        const size_t src = ((size_t)(ly - dy + 1) * We + (lx - dx + 1)) * (9 * C) + (size_t)(3 * r + sct) * C + 4 * c4;
        reinterpret_cast<float4*>(Vp)[idx] = *reinterpret_cast<const float4*>(Pc + src);
    }
}

// 16-bit modes: the same matrix transposed, VpT[blk*C + co][l] (bf16 or half, row stride ldt >= L, pad columns zeroed), the
// [N][K] operand of the 16-bit NT GEMM (gemm_h16.hip)
template <bool F16>
__global__ void csa_gather_vprime_t_h16_kernel(const float* __restrict__ Pc, int Hh, int Wh, int C, unsigned short* __restrict__ VpT,
                                                int ldt) {
    const long n = (long)25 * C * ldt;
    const int We = Wh + 3, L = Hh * Wh;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int l = (int)(idx % ldt);
        const int col = (int)(idx / ldt);
        float v = 0.f;
        if (l < L) {
            const int blk = col / C, co = col - blk * C;
            const int ly = l / Wh, lx = l - ly * Wh;
            int dy, dx, r, sct;
            if (blk < 16) { dy = blk / 4 - 2; dx = blk % 4 - 2; r = b3_subset(dy); sct = b3_subset(dx); }
            else if (blk < 20) { dy = 0; dx = blk - 16 - 2; r = 2; sct = b3_subset(dx); }
            else if (blk < 24) { dy = blk - 20 - 2; dx = 0; r = b3_subset(dy); sct = 2; }
            else { dy = 0; dx = 0; r = 2; sct = 2; }
            v = Pc[((size_t)(ly - dy + 1) * We + (lx - dx + 1)) * (9 * C) + (size_t)(3 * r + sct) * C + co];
        }
        VpT[idx] = to_h16<F16>(v);
    }
}

// out[(y',x')][co] = (bd[co] + sum over the 16 offsets of O'[(y'+dy, x'+dx)][blk*C + co] (variants on row/column 0)) / 6
__global__ void csa_gather_out_kernel(const float* __restrict__ Op, const float* __restrict__ Otop, const float* __restrict__ Oleft,
                                      const float* __restrict__ Otl, const float* __restrict__ bd, int H, int W, int Hp, int Wp,
                                      int C, float* __restrict__ out, int ld_out, long ld_main, long ld_top, long ld_left) {
    const int c4n = C >> 2;
    const long n = (long)H * W * c4n;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const long pix = idx / c4n;
        const int x = (int)(pix % W), y = (int)(pix / W);
        float4 acc = *reinterpret_cast<const float4*>(bd + 4 * c4);
        for (int dy = -2; dy <= 1; ++dy) {
            const int py = y + dy;
            if (py < 0 || py >= Hp) continue;
            for (int dx = -2; dx <= 1; ++dx) {
                const int px = x + dx;
                if (px < 0 || px >= Wp) continue;
                const float* src;
                if (dy == 0 && y == 0 && dx == 0 && x == 0) src = Otl + 4 * c4;
                else if (dy == 0 && y == 0) src = Otop + (size_t)px * ld_top + (size_t)(dx + 2) * C + 4 * c4;   // py == 0
                else if (dx == 0 && x == 0) src = Oleft + (size_t)py * ld_left + (size_t)(dy + 2) * C + 4 * c4; // px == 0
                else src = Op + ((size_t)py * Wp + px) * ld_main + (size_t)((dy + 2) * 4 + (dx + 2)) * C + 4 * c4;
                const float4 v = *reinterpret_cast<const float4*>(src);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        acc.x /= 6.f; acc.y /= 6.f; acc.z /= 6.f; acc.w /= 6.f;
        *reinterpret_cast<float4*>(out + (size_t)pix * ld_out + 4 * c4) = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// restorer plumbing
// ---------------------------------------------------------------------------------------------
struct Vec3 { float v[3]; };

__global__ void normalize_kernel(const float* __restrict__ lq, float* __restrict__ out, long HW, Vec3 mean, Vec3 std) {
    const long n = 3 * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i / HW);
        out[i] = (lq[i] - mean.v[c]) / std.v[c];
    }
}

__global__ void denorm_clamp_kernel(const float* __restrict__ pred, float* __restrict__ out, long HW, Vec3 mean, Vec3 std) {
    const long n = 3 * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i / HW);
        const long p = i - (long)c * HW;
        const float v = add_rn(mul_rn(pred[p * 3 + c], std.v[c]), mean.v[c]);
        out[i] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

__global__ void tile_blend_kernel(float* __restrict__ E, float* __restrict__ Wt, int Himg, int Wimg,
                                  const float* __restrict__ tile, int y0, int x0, int th, int tw) {
    const long n = 3L * th * tw;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i / ((long)th * tw));
        const long p = i - (long)c * th * tw;
        const int y = (int)(p / tw), x = (int)(p % tw);
        const size_t o = ((size_t)c * Himg + (y0 + y)) * Wimg + (x0 + x);
        E[o] += tile[p * 3 + c];
        Wt[o] += 1.f;
    }
}

__global__ void tile_finalize_kernel(const float* __restrict__ E, const float* __restrict__ Wt,
                                     float* __restrict__ out, long HW) {
    const long n = 3 * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i / HW);
        const long p = i - (long)c * HW;
        out[p * 3 + c] = E[i] / Wt[i];
    }
}

static inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// ---- host wrappers (used by the composite ops as well) ----------------------------------------
int nchw_to_hwc(const float* src, float* dst, int C, int H, int W, int ld, hipStream_t s) {
    ProfScope prof("nchw_to_hwc", s);
    dim3 grid(ceil_div((long)H * W, 32), ceil_div(C, 32));
    hipLaunchKernelGGL(nchw_to_hwc_kernel, grid, dim3(256), 0, s, src, dst, C, H * W, ld);
    return launch_status("nchw_to_hwc");
}

int hwc_to_nchw(const float* src, int ld, float* dst, int C, int H, int W, hipStream_t s) {
    ProfScope prof("hwc_to_nchw", s);
    dim3 grid(ceil_div((long)H * W, 32), ceil_div(C, 32));
    hipLaunchKernelGGL(hwc_to_nchw_kernel, grid, dim3(256), 0, s, src, ld, dst, C, H * W);
    return launch_status("hwc_to_nchw");
}

int pad_reflect(const float* src, int ld_src, int H, int W, int C, float* dst, int Hp, int Wp, hipStream_t s) {
    ProfScope prof("pad_reflect", s);
    hipLaunchKernelGGL(pad_reflect_kernel, dim3(ew_grid((long)Hp * Wp * C / 4)), dim3(256), 0, s, src, ld_src, H,
                       W, C, dst, Hp, Wp);
    return launch_status("pad_reflect");
}

int avgpool2(const float* src, int Hp, int Wp, int C, float* dst, hipStream_t s) {
    ProfScope prof("avgpool2", s);
    hipLaunchKernelGGL(avgpool2_kernel, dim3(ew_grid((long)Hp * Wp * C / 16)), dim3(256), 0, s, src, Hp, Wp, C, dst);
    return launch_status("avgpool2");
}

int patch_rows(const float* src, int ld_src, int Hs, int Ws, int Cs, int k, int stride, int pad, int OH, int OW,
               float* out, int ld_out, int normalize, float floor_, hipStream_t s, const char* tag) {
    CIAOSR_CHECK_ARG((Cs & 3) == 0 && (ld_src & 3) == 0 && (ld_out & 3) == 0);
    CIAOSR_CHECK_ARG(aligned16(src) && aligned16(out));
    PatchP p{src, out, ld_src, Hs, Ws, Cs, k, stride, pad, OH, OW, ld_out, normalize, floor_};
    ProfScope prof(tag ? tag : "patch_rows", s);
    hipLaunchKernelGGL(patch_rows_kernel, dim3(ceil_div((long)OH * OW, 4)), dim3(256), 0, s, p);
    return launch_status("patch_rows");
}

// Row statistics only: stats[row] = (row max x log2 e, 1 / sum of exp2(v log2 e - max log2 e)) -- what the attn.V GEMM's operand
// staging (gemm_f32.hip) needs to form the probabilities on the fly with one FMA, one v_exp_f32 and one multiply per element; the
// sum uses the same expression, so every row still sums to 1 to rounding.  The in-place rewrite of S (one 1.36-GB write and one
// read per 192x192 tile) disappears.  Single pass, the whole row in registers when it fits (rows of up to 64 x SMV float4).
constexpr float kLog2eP = 1.4426950408889634f;
__global__ __launch_bounds__(256) void softmax_stats_kernel(const float* __restrict__ S, long rows, int L, int ld, float2* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + (size_t)row * ld;
    const int n4 = ld >> 2;
    float m = -INFINITY;
    for (int t = lane; t < n4; t += 64) {
        const float4 v = reinterpret_cast<const float4*>(s)[t];
        const int c = 4 * t;
        if (c < L) m = fmaxf(m, v.x);
        if (c + 1 < L) m = fmaxf(m, v.y);
        if (c + 2 < L) m = fmaxf(m, v.z);
        if (c + 3 < L) m = fmaxf(m, v.w);
    }
    m = wave_max(m) * kLog2eP;
    float sum = 0.f;
    for (int t = lane; t < n4; t += 64) {
        const float4 v = reinterpret_cast<const float4*>(s)[t];
        const int c = 4 * t;
        if (c < L) sum += __builtin_amdgcn_exp2f(__builtin_fmaf(v.x, kLog2eP, -m));
        if (c + 1 < L) sum += __builtin_amdgcn_exp2f(__builtin_fmaf(v.y, kLog2eP, -m));
        if (c + 2 < L) sum += __builtin_amdgcn_exp2f(__builtin_fmaf(v.z, kLog2eP, -m));
        if (c + 3 < L) sum += __builtin_amdgcn_exp2f(__builtin_fmaf(v.w, kLog2eP, -m));
    }
    sum = wave_sum(sum);
    if (lane == 0) stats[row] = make_float2(m, 1.f / sum);
}
__global__ __launch_bounds__(256) void softmax_stats_reg_kernel(const float* __restrict__ S, long rows, int L, int ld, float2* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* s = reinterpret_cast<const float4*>(S + (size_t)row * ld);
    const int n4 = ld >> 2;
    float4 v[SMV];
#pragma unroll
    for (int i = 0; i < SMV; ++i) {
        const int t = lane + 64 * i, c = 4 * t;
        v[i] = (t < n4 && c < L) ? s[t] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (c + 1 >= L) v[i].y = -INFINITY;
        if (c + 2 >= L) v[i].z = -INFINITY;
        if (c + 3 >= L) v[i].w = -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < SMV; ++i) m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    m = wave_max(m) * kLog2eP;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SMV; ++i)                        // exp2(-inf) = 0 for the masked tail
        sum += (__builtin_amdgcn_exp2f(__builtin_fmaf(v[i].x, kLog2eP, -m)) + __builtin_amdgcn_exp2f(__builtin_fmaf(v[i].y, kLog2eP, -m))) +
               (__builtin_amdgcn_exp2f(__builtin_fmaf(v[i].z, kLog2eP, -m)) + __builtin_amdgcn_exp2f(__builtin_fmaf(v[i].w, kLog2eP, -m)));
    sum = wave_sum(sum);
    if (lane == 0) stats[row] = make_float2(m, 1.f / sum);
}

int softmax_stats_rows(const float* S, long rows, int L, int ld, float* stats2, hipStream_t s) {
    ProfScope prof("softmax_stats", s);
    float2* st = reinterpret_cast<float2*>(stats2);
    if ((ld >> 2) <= 64 * SMV && (ld >> 2) > 64 * 8)
        hipLaunchKernelGGL(softmax_stats_reg_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, S, rows, L, ld, st);
    else
        hipLaunchKernelGGL(softmax_stats_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, S, rows, L, ld, st);
    return launch_status("softmax_stats");
}

int softmax_rows(float* S, long rows, int L, int ld, hipStream_t s) {
    ProfScope prof("softmax_rows", s);
    if ((ld >> 2) <= 64 * SMV && (ld >> 2) > 64 * 8)       // long rows that still fit the register file: one pass
        hipLaunchKernelGGL(softmax_rows_reg_kernel<0>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, S, rows, L, ld, nullptr, 0);
    else
        hipLaunchKernelGGL(softmax_rows_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, S, rows, L, ld);
    return launch_status("softmax_rows");
}

// fp32 logits -> bf16 / half probabilities, single pass; false when the row does not fit the register file
bool softmax_rows_reg_h16(const float* S, long rows, int L, int ld, unsigned short* P, int ldp, bool f16, hipStream_t s) {
    if ((ldp >> 2) > 64 * SMV) return false;
    if (f16)
        hipLaunchKernelGGL(softmax_rows_reg_kernel<2>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, const_cast<float*>(S), rows, L, ld, P, ldp);
    else
        hipLaunchKernelGGL(softmax_rows_reg_kernel<1>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, const_cast<float*>(S), rows, L, ld, P, ldp);
    return true;
}

int csa_gather_vprime(const float* Pc, int Hh, int Wh, int C, float* Vp, hipStream_t s) {
    ProfScope prof("csa_gather_vprime", s);
    hipLaunchKernelGGL(csa_gather_vprime_kernel, dim3(ew_grid((long)Hh * Wh * 25 * C / 4)), dim3(256), 0, s, Pc, Hh, Wh, C, Vp);
    return launch_status("csa_gather_vprime");
}

int csa_gather_out(const float* Op, const float* Otop, const float* Oleft, const float* Otl, const float* bd, int H, int W,
                   int Hp, int Wp, int C, float* out, int ld_out, long ld_main, long ld_top, long ld_left, hipStream_t s) {
    ProfScope prof("csa_gather_out", s);
    hipLaunchKernelGGL(csa_gather_out_kernel, dim3(ew_grid((long)H * W * C / 4)), dim3(256), 0, s, Op, Otop, Oleft, Otl, bd, H,
                       W, Hp, Wp, C, out, ld_out, ld_main, ld_top, ld_left);
    return launch_status("csa_gather_out");
}

int csa_gather_vprime_t_h16(const float* Pc, int Hh, int Wh, int C, unsigned short* VpT, int ldt, bool f16, hipStream_t s) {
    ProfScope prof("csa_gather_vprime", s);
    if (f16)
        hipLaunchKernelGGL(csa_gather_vprime_t_h16_kernel<true>, dim3(ew_grid((long)25 * C * ldt)), dim3(256), 0, s, Pc, Hh, Wh, C, VpT, ldt);
    else
        hipLaunchKernelGGL(csa_gather_vprime_t_h16_kernel<false>, dim3(ew_grid((long)25 * C * ldt)), dim3(256), 0, s, Pc, Hh, Wh, C, VpT, ldt);
    return launch_status("csa_gather_vprime_t");
}

int downsample(const float* src, int Hp, int Wp, int C, int scale, float* dst, hipStream_t s) {
    CIAOSR_CHECK_ARG(scale >= 2 && scale <= 4 && Hp % scale == 0 && Wp % scale == 0);
    ProfScope prof("avgpool2", s);
    hipLaunchKernelGGL(downsample_kernel, dim3(ew_grid((long)(Hp / scale) * (Wp / scale) * C / 4)), dim3(256), 0, s, src, Hp, Wp, C, scale,
                       dst);
    return launch_status("downsample");
}

int fold_s(const float* O, int ldo, int Hp, int Wp, int C, int scale, float* Y, hipStream_t s) {
    ProfScope prof("fold_gather", s);
    hipLaunchKernelGGL(fold_s_kernel, dim3(ew_grid((long)scale * scale * Hp * Wp * C / 4)), dim3(256), 0, s, O, ldo, Hp, Wp, C, scale, Y);
    return launch_status("fold_s");
}

int fold(const float* O, int ldo, int Hp, int Wp, int C, float* Y, hipStream_t s) {
    ProfScope prof("fold_gather", s);
    hipLaunchKernelGGL(fold_kernel, dim3(ew_grid((long)Hp * Wp * C)), dim3(256), 0, s, O, ldo, Hp, Wp, C, Y);
    return launch_status("fold");
}

}  // namespace ciaosr

using namespace ciaosr;

extern "C" int ciaosr_nchw_to_hwc_f32(const float* src, float* dst, int C, int H, int W, int ld_dst, void* stream) {
    CIAOSR_CHECK_ARG(src && dst && C > 0 && H > 0 && W > 0 && ld_dst >= C);
    return nchw_to_hwc(src, dst, C, H, W, ld_dst, (hipStream_t)stream);
}

extern "C" int ciaosr_hwc_to_nchw_f32(const float* src, int ld_src, float* dst, int C, int H, int W, void* stream) {
    CIAOSR_CHECK_ARG(src && dst && C > 0 && H > 0 && W > 0 && ld_src >= C);
    return hwc_to_nchw(src, ld_src, dst, C, H, W, (hipStream_t)stream);
}

extern "C" int ciaosr_patch_rows_f32(const float* src_hwc, int ld_src, int Hs, int Ws, int Cs, int ksize,
                                     int stride, int pad, int OH, int OW, float* out, int ld_out,
                                     int l2_normalize, float norm_floor, void* stream) {
    CIAOSR_CHECK_ARG(src_hwc && out && ksize > 0 && stride > 0 && OH > 0 && OW > 0);
    CIAOSR_CHECK_ARG(ld_out >= ksize * ksize * Cs && ld_src >= Cs);
    return patch_rows(src_hwc, ld_src, Hs, Ws, Cs, ksize, stride, pad, OH, OW, out, ld_out, l2_normalize,
                      norm_floor, (hipStream_t)stream, nullptr);
}

extern "C" int ciaosr_normalize_f32(const float* lq, float* out, int H, int W, const float* mean3,
                                    const float* std3, void* stream) {
    CIAOSR_CHECK_ARG(lq && out && mean3 && std3 && H > 0 && W > 0);
    Vec3 m{{mean3[0], mean3[1], mean3[2]}}, sd{{std3[0], std3[1], std3[2]}};
    ProfScope prof("normalize", (hipStream_t)stream);
    hipLaunchKernelGGL(normalize_kernel, dim3(ew_grid(3L * H * W)), dim3(256), 0, (hipStream_t)stream, lq, out,
                       (long)H * W, m, sd);
    return launch_status("normalize");
}

extern "C" int ciaosr_denorm_clamp_f32(const float* pred_q3, float* out_chw, int H, int W, const float* mean3,
                                       const float* std3, void* stream) {
    CIAOSR_CHECK_ARG(pred_q3 && out_chw && mean3 && std3 && H > 0 && W > 0);
    Vec3 m{{mean3[0], mean3[1], mean3[2]}}, sd{{std3[0], std3[1], std3[2]}};
    ProfScope prof("denorm_clamp", (hipStream_t)stream);
    hipLaunchKernelGGL(denorm_clamp_kernel, dim3(ew_grid(3L * H * W)), dim3(256), 0, (hipStream_t)stream, pred_q3,
                       out_chw, (long)H * W, m, sd);
    return launch_status("denorm_clamp");
}

extern "C" int ciaosr_tile_blend_f32(float* E, float* Wt, int Himg, int Wimg, const float* tile_q3, int y0,
                                     int x0, int th, int tw, void* stream) {
    CIAOSR_CHECK_ARG(E && Wt && tile_q3 && y0 >= 0 && x0 >= 0 && y0 + th <= Himg && x0 + tw <= Wimg);
    ProfScope prof("tile_blend", (hipStream_t)stream);
    hipLaunchKernelGGL(tile_blend_kernel, dim3(ew_grid(3L * th * tw)), dim3(256), 0, (hipStream_t)stream, E, Wt,
                       Himg, Wimg, tile_q3, y0, x0, th, tw);
    return launch_status("tile_blend");
}

extern "C" int ciaosr_tile_finalize_f32(const float* E, const float* Wt, float* out_q3, int Himg, int Wimg,
                                        void* stream) {
    CIAOSR_CHECK_ARG(E && Wt && out_q3 && Himg > 0 && Wimg > 0);
    ProfScope prof("tile_finalize", (hipStream_t)stream);
    hipLaunchKernelGGL(tile_finalize_kernel, dim3(ew_grid(3L * Himg * Wimg)), dim3(256), 0, (hipStream_t)stream, E,
                       Wt, out_q3, (long)Himg * Wimg);
    return launch_status("tile_finalize");
}
