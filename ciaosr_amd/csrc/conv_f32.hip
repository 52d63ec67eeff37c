// Implicit-GEMM convolution (1x1 / 3x3, stride 1, zero 'same' padding) on channels-last maps,
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), with split-K for small images.
//
//   dst[p][co] = act((sum_{tap,ci} src[p + tap][ci] * w[co][tap*Cin + ci] + bias[co]) * alpha) + res[p][co]
//
// The A operand is never materialised: a K-tile of 32 input channels lies inside one 3x3 tap, so the
// tile load is 64 pixels x 128 contiguous bytes of the (shifted) channels-last map.  Dense-block
// concatenation (RDN) is free: every layer writes its output channels into the next columns of one
// [HW][C_total] buffer (ld_dst) and the next layer reads a wider prefix of it (ld_src).
// Tile 64x64x32, 4 waves as 2x2 of one 32x32 MFMA tile each.  With only H*W/64 x Cout/64 tiles
// (36 at 48x48) the K loop is split over blockIdx.y and a second kernel reduces the partial slabs in
// a fixed order (deterministic) and applies bias / activation / residual.
//
// Replaces the encoder trunk convolutions the reference runs through torch conv2d:
// RDN / EDSR `gen_feature` (ciaosr_net.py:321-342, :393-408).
#include "ops.h"

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CBM = 64, CBN = 64, CBK = 32, CLD = CBK + 4;

struct ConvP {
    const float* src; int ld_src; int H, W, Cin;
    const float* wgt; int ldw;
    const float* bias;
    int Cout, taps;
    float* dst; int ld_dst;
    float* dst2; int ld_dst2;
    const float* res; int ld_res;
    int act; float alpha;
    int M, K;
    int tiles_n, splitk, kt_per_split;
    float* partial;   // [splitk][M][Cout] when splitk > 1
};

__device__ __forceinline__ void conv_epilogue(const ConvP& p, int row, int col, float v) {
    v = (v + (p.bias ? p.bias[col] : 0.f)) * p.alpha;
    if (p.act == CIAOSR_ACT_RELU) v = fmaxf(v, 0.f);
    if (p.res) v += p.res[(size_t)row * p.ld_res + col];
    p.dst[(size_t)row * p.ld_dst + col] = v;
    if (p.dst2) p.dst2[(size_t)row * p.ld_dst2 + col] = v;
}

__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvP p) {
    __shared__ __attribute__((aligned(16))) float As[2][CBM * CLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][CBN * CLD];
    const int tile = blockIdx.x;
    const int m0 = (tile / p.tiles_n) * CBM, n0 = (tile % p.tiles_n) * CBN;
    const int split = blockIdx.y;
    const int nk_total = p.K / CBK;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(nk_total, kt0 + p.kt_per_split);

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;

    // this thread stages rows r0 and r0+32 (A: pixels, B: output channels), float4 column c4
    const int r0 = t >> 3, c4 = (t & 7) * 4;
    int py[2], px[2];
    bool pv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int m = m0 + r0 + 32 * s;
        pv[s] = m < p.M;
        py[s] = pv[s] ? m / p.W : 0;
        px[s] = pv[s] ? m - py[s] * p.W : 0;
    }
    float4 ra[2], rb[2];
    auto load_tiles = [&](int kt) {
        const int k0 = kt * CBK;
        const int tap = k0 / p.Cin, cc = k0 - tap * p.Cin + c4;
        const int dy = p.taps == 9 ? tap / 3 - 1 : 0, dx = p.taps == 9 ? tap % 3 - 1 : 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int y = py[s] + dy, x = px[s] + dx;
            const bool ok = pv[s] && y >= 0 && y < p.H && x >= 0 && x < p.W;
            ra[s] = ok ? *reinterpret_cast<const float4*>(p.src + ((size_t)y * p.W + x) * p.ld_src + cc)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
            const int n = n0 + r0 + 32 * s;
            rb[s] = n < p.Cout ? *reinterpret_cast<const float4*>(p.wgt + (size_t)n * p.ldw + k0 + c4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            *reinterpret_cast<float4*>(&As[buf][(r0 + 32 * s) * CLD + c4]) = ra[s];
            *reinterpret_cast<float4*>(&Bs[buf][(r0 + 32 * s) * CLD + c4]) = rb[s];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if (kt0 < kt1) {
        load_tiles(kt0);
        store_tiles(0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < kt1) load_tiles(kt + 1);
        const float* a = &As[cur][(wm * 32 + li) * CLD + 4 * lh];
        const float* b = &Bs[cur][(wn * 32 + li) * CLD + 4 * lh];
#pragma unroll
        for (int j = 0; j < CBK / 8; ++j) {
            const float4 fa = *reinterpret_cast<const float4*>(a + 8 * j);
            const float4 fb = *reinterpret_cast<const float4*>(b + 8 * j);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
        }
        if (kt + 1 < kt1) store_tiles(cur ^ 1);
        __syncthreads();
    }

    const int col = n0 + wn * 32 + li;
    if (col >= p.Cout) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= p.M) continue;
        if (p.splitk > 1)
            p.partial[((size_t)split * p.M + row) * p.Cout + col] = acc[r];
        else
            conv_epilogue(p, row, col, acc[r]);
    }
}

__global__ void conv_reduce_kernel(ConvP p) {
    const long n = (long)p.M * p.Cout;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += p.partial[(size_t)s * n + i];   // fixed order: deterministic
        const int row = (int)(i / p.Cout), col = (int)(i - (long)row * p.Cout);
        conv_epilogue(p, row, col, v);
    }
}

size_t conv_partial_floats(int H, int W, int Cin, int Cout, int taps) {
    // worst case split count is bounded by 16
    return (size_t)16 * H * W * Cout;
}

// src [H*W][ld_src] (first Cin columns) -> dst [H*W][ld_dst] (first Cout columns)
int conv2d_hwc(const float* src, int ld_src, int H, int W, int Cin, const float* wgt, int ldw, const float* bias,
               int Cout, int ksize, float* dst, int ld_dst, float* dst2, int ld_dst2, const float* res, int ld_res,
               int act, float alpha, float* partial, size_t partial_floats, hipStream_t s, const char* tag) {
    CIAOSR_CHECK_ARG(src && wgt && dst && (ksize == 1 || ksize == 3));
    CIAOSR_CHECK_ARG(Cin % CBK == 0 && (ld_src & 3) == 0 && (ldw & 3) == 0);
    CIAOSR_CHECK_ARG(aligned16(src) && aligned16(wgt));
    ConvP p;
    p.src = src; p.ld_src = ld_src; p.H = H; p.W = W; p.Cin = Cin;
    p.wgt = wgt; p.ldw = ldw; p.bias = bias; p.Cout = Cout; p.taps = ksize * ksize;
    p.dst = dst; p.ld_dst = ld_dst; p.dst2 = dst2; p.ld_dst2 = ld_dst2; p.res = res; p.ld_res = ld_res;
    p.act = act; p.alpha = alpha;
    p.M = H * W; p.K = p.taps * Cin;
    p.tiles_n = ceil_div(Cout, CBN);
    const int tiles = ceil_div(p.M, CBM) * p.tiles_n;
    const int nk = p.K / CBK;
    // enough workgroups for 256 CUs x 2: split the K loop when the tile grid is small
    int splitk = 1;
    if (tiles < 384 && partial) {
        splitk = ceil_div(512, tiles);
        if (splitk > nk) splitk = nk;
        if (splitk > 16) splitk = 16;
        if ((size_t)splitk * p.M * Cout > partial_floats) splitk = (int)(partial_floats / ((size_t)p.M * Cout));
        if (splitk < 1) splitk = 1;
    }
    p.kt_per_split = ceil_div(nk, splitk);
    splitk = ceil_div(nk, p.kt_per_split);
    p.splitk = splitk;
    p.partial = partial;
    {
        ProfScope prof(tag ? tag : (ksize == 3 ? "conv3x3" : "conv1x1"), s);
        hipLaunchKernelGGL(conv_gemm_kernel, dim3(tiles, splitk), dim3(256), 0, s, p);
    }
    int rc = launch_status("conv_gemm");
    if (rc != CIAOSR_OK) return rc;
    if (splitk > 1) {
        ProfScope prof("conv_splitk_reduce", s);
        long n = (long)p.M * Cout;
        int grid = (int)((n + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(conv_reduce_kernel, dim3(grid), dim3(256), 0, s, p);
        rc = launch_status("conv_reduce");
    }
    return rc;
}

}  // namespace ciaosr
