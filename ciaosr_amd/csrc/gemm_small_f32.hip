// Small-problem NT GEMM, exact-fp32 MFMA: C[M][N] = act(A[M][K] . W[N][K]^T + bias) (+ res), for the many launches of
// this path whose whole problem is a few MFLOP -- the Linear layers of the SwinIR trunk (swinir_net.py:114-146, :23-36
// at 2304 tokens), the LFF 1x1 convolution of an RDN block on a 48x48 map, the head's per-LR-pixel tables.  There the
// 128x128-tile GEMM (gemm_f32.hip) has 36 workgroups and the tap-major convolution pays LDS stages and barriers; a
// launch is a latency problem, so this kernel has NO staging at all:
//   * workgroup = 64 (or 32, when that is needed to cover the chip) rows x 32 columns, 4 waves that split K (wave w owns a contiguous quarter of the 8-deep k-chunks);
//   * both MFMA operands are read straight from global/L2 into registers in fragment order (a lane's float4 =
//     4 consecutive k of one row; the consistent k permutation of gemm_f32.hip makes it feed 4 MFMAs), two chunks
//     ahead of their use; no LDS, no barrier in the main loop;
//   * one fixed-order K-slice reduction through 32 KB of LDS, then a float4 epilogue (bias, ReLU / exact GELU,
//     residual, optional second destination).
// Out-of-range rows / columns / k go through buffer descriptors (loads return 0, stores vanish).  K % 4 == 0.
#include "ops.h"

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kOobGS = 0xFFFFFFF0u;
constexpr size_t kGemmSmallLds = 32768;      // K-slice reduction scratch: 4 waves x 2 tiles x 4 quads x 64 lanes x 16 B

struct GemmSmallP {
    const float* A; int lda; unsigned a_bytes;
    const float* W; int ldw; unsigned w_bytes;
    const float* bias;
    float* C; int ldc; unsigned c_bytes;
    float* C2; int ldc2; unsigned c2_bytes;
    const float* res; int ldres; unsigned res_bytes;
    int M, N, K, tiles_n, act;
    float slope, alpha;      // out = act((acc + bias) * alpha) + res
};

__device__ __forceinline__ float4 gs_load4(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ void gs_store4(__amdgpu_buffer_rsrc_t rs, unsigned off, float4 v) {
    i32x4 iv;
    iv.x = __float_as_int(v.x); iv.y = __float_as_int(v.y); iv.z = __float_as_int(v.z); iv.w = __float_as_int(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(iv, rs, (int)off, 0, 0);
}

template <int MT>      // 32-row tiles per workgroup: 2 (64 rows) or 1 (32 rows, for problems with few workgroups)
__global__ __launch_bounds__(256) void gemm_small_f32_kernel(GemmSmallP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsg[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
    const int m0 = (blockIdx.x / p.tiles_n) * (32 * MT), n0 = (blockIdx.x % p.tiles_n) * 32;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, p.w_bytes, 0x00020000);

    // this wave's K range: chunks [c0, c0 + cpw) of 8 consecutive k; a lane reads k = 8c + 4lh .. + 3
    const int cpw = (((p.K + 7) >> 3) + 3) >> 2;   // ceil(ceil(K / 8) / 4); K % 4 == 0, loads past K read zeros
    const int c0 = w * cpw;
    const int kl = 8 * c0 + 4 * lh;           // this lane's first k
    unsigned aoff[MT], woff;
#pragma unroll
    for (int r = 0; r < MT; ++r) {
        const int m = m0 + 32 * r + li;
        aoff[r] = m < p.M ? ((unsigned)m * (unsigned)p.lda + (unsigned)(8 * c0 + 4 * lh)) * 4u : kOobGS;
    }
    {
        const int n = n0 + li;
        woff = n < p.N ? ((unsigned)n * (unsigned)p.ldw + (unsigned)(8 * c0 + 4 * lh)) * 4u : kOobGS;
    }
    auto ld_a = [&](int r, int c) { return gs_load4(rs_a, (aoff[r] == kOobGS || kl + 8 * c >= p.K) ? kOobGS : aoff[r] + (unsigned)c * 32u); };
    auto ld_w = [&](int c) { return gs_load4(rs_w, (woff == kOobGS || kl + 8 * c >= p.K) ? kOobGS : woff + (unsigned)c * 32u); };

    f32x16 acc[MT];
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    // two chunks in flight ahead of the one being multiplied
    float4 a0[MT], a1[MT], a2[MT], b0, b1, b2;
#pragma unroll
    for (int r = 0; r < MT; ++r) a0[r] = ld_a(r, 0);
    b0 = ld_w(0);
#pragma unroll
    for (int r = 0; r < MT; ++r) a1[r] = a0[r];
    b1 = b0;
    if (cpw > 1) {
#pragma unroll
        for (int r = 0; r < MT; ++r) a1[r] = ld_a(r, 1);
        b1 = ld_w(1);
    }
#pragma unroll
    for (int r = 0; r < MT; ++r) a2[r] = a1[r];
    b2 = b1;
#pragma unroll 1
    for (int c = 0; c < cpw; ++c) {
        if (c + 2 < cpw) {
#pragma unroll
            for (int r = 0; r < MT; ++r) a2[r] = ld_a(r, c + 2);
            b2 = ld_w(c + 2);
        }
#pragma unroll
        for (int r = 0; r < MT; ++r) {
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.x, a0[r].x, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.y, a0[r].y, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.z, a0[r].z, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.w, a0[r].w, acc[r], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < MT; ++r) { a0[r] = a1[r]; a1[r] = a2[r]; }
        b0 = b1; b1 = b2;
    }

    // K-slice reduction (swapped operands: a lane owns row li of tile r and columns 8q + 4lh .. +3 per register quad)
    float4* red = reinterpret_cast<float4*>(ldsg);
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            red[((w * MT + r) * 4 + q) * 64 + lane] = make_float4(acc[r][4 * q], acc[r][4 * q + 1], acc[r][4 * q + 2], acc[r][4 * q + 3]);
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, p.c_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c2 = __builtin_amdgcn_make_buffer_rsrc(p.C2 ? p.C2 : p.C, 0, p.C2 ? p.c2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.A), 0, p.res ? p.res_bytes : 0u, 0x00020000);
    // bias and residual of every unit first (in flight together, under the LDS reduction): inside the unit loop, behind the stores
    // they may alias, each would be its own round trip -- a tenth of a ~9-us launch
    float4 bq[MT], rq[MT];
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        const int unit = t + 256 * u;
        const int ul = unit & 63, q = (unit >> 6) & 3, r = unit >> 8;
        const int m = m0 + 32 * r + (ul & 31), n = n0 + 8 * q + 4 * (ul >> 5);
        const bool ok = m < p.M && n < p.N;
        bq[u] = (p.bias && ok) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        rq[u] = p.res ? gs_load4(rs_r, ok ? ((unsigned)m * (unsigned)p.ldres + (unsigned)n) * 4u : kOobGS) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        const int unit = t + 256 * u;
        const int ul = unit & 63, q = (unit >> 6) & 3, r = unit >> 8;
        float4 v = red[((0 * MT + r) * 4 + q) * 64 + ul];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const float4 o = red[((ww * MT + r) * 4 + q) * 64 + ul];
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        const int m = m0 + 32 * r + (ul & 31), n = n0 + 8 * q + 4 * (ul >> 5);
        const bool ok = m < p.M && n < p.N;                       // N % 4 == 0
        {
            const float4 b = bq[u];
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
        if (p.act == CIAOSR_ACT_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if (p.act == CIAOSR_ACT_PRELU) {
            v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope;
            v.z = v.z > 0.f ? v.z : v.z * p.slope; v.w = v.w > 0.f ? v.w : v.w * p.slope;
        } else if (p.act == CIAOSR_ACT_GELU) {
            v.x = 0.5f * v.x * (1.f + erff(v.x * 0.70710678118654752f));
            v.y = 0.5f * v.y * (1.f + erff(v.y * 0.70710678118654752f));
            v.z = 0.5f * v.z * (1.f + erff(v.z * 0.70710678118654752f));
            v.w = 0.5f * v.w * (1.f + erff(v.w * 0.70710678118654752f));
        }
        if (p.res) {
            const float4 rr = rq[u];
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        gs_store4(rs_c, ok ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 4u : kOobGS, v);
        if (p.C2) gs_store4(rs_c2, ok ? ((unsigned)m * (unsigned)p.ldc2 + (unsigned)n) * 4u : kOobGS, v);
    }
}

// true when the problem fits this kernel's envelope (small M, K a multiple of 4, 16-byte aligned rows)
bool gemm_small_ok(int M, int N, int K, int lda, int ldw) {
    return M > 0 && M <= 16384 && N > 0 && (N & 3) == 0 && K >= 32 && (K & 3) == 0 && (lda & 3) == 0 && (ldw & 3) == 0;
}

int gemm_small_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, float* C2, int ldc2,
                   const float* res, int ldres, int M, int N, int K, int act, float slope, float alpha, hipStream_t s, const char* tag) {
    CIAOSR_CHECK_ARG(A && W && C && gemm_small_ok(M, N, K, lda, ldw) && (ldc & 3) == 0 && (ldc2 & 3) == 0 && (ldres & 3) == 0);
    CIAOSR_CHECK_ARG(aligned16(A) && aligned16(W) && aligned16(C) && (!C2 || aligned16(C2)) && (!res || aligned16(res)) &&
                     (!bias || aligned16(bias)));
    GemmSmallP p;
    const size_t ab = ((size_t)(M - 1) * lda + K) * 4, wb = ((size_t)(N - 1) * ldw + K) * 4, cb = ((size_t)(M - 1) * ldc + N) * 4;
    const size_t c2b = C2 ? ((size_t)(M - 1) * ldc2 + N) * 4 : 0, rb = res ? ((size_t)(M - 1) * ldres + N) * 4 : 0;
    CIAOSR_CHECK_ARG(ab < 0xFFFFFF00ull && wb < 0xFFFFFF00ull && cb < 0xFFFFFF00ull && c2b < 0xFFFFFF00ull && rb < 0xFFFFFF00ull);
    p.A = A; p.lda = lda; p.a_bytes = (unsigned)ab;
    p.W = W; p.ldw = ldw; p.w_bytes = (unsigned)wb;
    p.bias = bias;
    p.C = C; p.ldc = ldc; p.c_bytes = (unsigned)cb;
    p.C2 = C2; p.ldc2 = ldc2; p.c2_bytes = (unsigned)c2b;
    p.res = res; p.ldres = ldres; p.res_bytes = (unsigned)rb;
    p.M = M; p.N = N; p.K = K; p.tiles_n = ceil_div(N, 32); p.act = act; p.slope = slope; p.alpha = alpha;
    ProfScope prof(tag ? tag : "gemm_small_f32", s);
    if (ceil_div(M, 64) * p.tiles_n < 192)        // too few 64-row workgroups to cover the chip: 32-row tiles
        hipLaunchKernelGGL(gemm_small_f32_kernel<1>, dim3(ceil_div(M, 32) * p.tiles_n), dim3(256), kGemmSmallLds, s, p);
    else
        hipLaunchKernelGGL(gemm_small_f32_kernel<2>, dim3(ceil_div(M, 64) * p.tiles_n), dim3(256), kGemmSmallLds, s, p);
    return launch_status("gemm_small_f32");
}

}  // namespace ciaosr
