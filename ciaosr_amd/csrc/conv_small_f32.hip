// 3x3 convolution (stride 1, zero 'same' padding) for SMALL maps, exact-fp32 MFMA: the no-stage design of
// dense_scatter_f32.hip with a general epilogue, for the trunk convolutions that are not dense-block steps -- EDSR
// residual blocks (ciaosr_net.py:393-408), RDN sfe2 / gff.1 (:321-342), the SwinIR RSTB and conv_after_body convolutions
// (swinir_net.py:449-459, :777).
//
//   dst[p][co] = act((sum_{tap,ci} src[p + tap][ci] * w[co][tap*Cin + ci] + bias[co]) * alpha) + res[p][co]
//
// Workgroup = 8x8 pixels x 32 output channels, 4 waves that split K (wave w owns channels 16w..16w+15 of every
// 64-channel input group).  Per input group: the 18 weight fragments of the wave (ciaosr_pack_fragments_f32 order)
// and the next 10x10 halo patch are requested from L2 while the current group's 144 MFMAs run out of the LDS patch;
// no staging of weights, one barrier pair per group.  One fixed-order K-slice reduction through LDS, float4 epilogue.
// The generic tap-major kernel of conv_f32.hip needs a split-K launch plus a reduce launch for these 72-workgroup
// layers (9 + 8 us); this is one launch.
#include "ops.h"

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int CT = 8, CP = CT + 2, CPS = 272;
constexpr int CPATCH = CP * CP * CPS;        // 27 200 B
constexpr size_t kConvSmallLds = 32768;
static_assert(CPATCH <= 32768, "the halo patch must fit the reduction scratch");
constexpr unsigned kOobC = 0xFFFFFFF0u;

struct ConvSmallP {
    const float* src; int ld_src; unsigned src_bytes;
    int H, W, tiles_x, n32, groups;          // groups = Cin / 64
    const float4* wf; int nj;                // fragments [n32][nj = 9*Cin/8][64 lanes]
    const float* bias; int Cout;
    float* dst; int ld_dst; unsigned dst_bytes;
    float* dst2; int ld_dst2; unsigned dst2_bytes;
    const float* res; int ld_res; unsigned res_bytes;
    int act; float alpha;
};

template <int MT>      // 32-pixel MFMA row tiles per workgroup: 2 (8x8 pixels) or 1 (8 wide x 4 high)
__global__ __launch_bounds__(256) void conv3x3_small_kernel(ConvSmallP p) {
    constexpr int TY = 4 * MT, PY = TY + 2;
    constexpr int NCH = PY * CP * 16, NLD = (NCH + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsc[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
    const int tile = blockIdx.x / p.n32, nt = blockIdx.x - tile * p.n32;
    const int ty0 = (tile / p.tiles_x) * TY, tx0 = (tile % p.tiles_x) * CT;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);

    unsigned goff[NLD];
#pragma unroll
    for (int s = 0; s < NLD; ++s) {
        const int c = t + 256 * s;
        const int px = c >> 4, part = c & 15;
        const int py = px / CP, pxx = px - py * CP;
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + pxx;
        const bool ok = c < NCH && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        goff[s] = ok ? ((unsigned)(gy * p.W + gx) * (unsigned)p.ld_src + (unsigned)part * 4u) * 4u : kOobC;
    }
    i32x4 P[NLD];
    auto load_patch = [&](int g) {
#pragma unroll
        for (int s = 0; s < NLD; ++s)
            P[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[s] == kOobC ? (int)kOobC : (int)(goff[s] + (unsigned)g * 256u), 0, 0);
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int s = 0; s < NLD; ++s) {
            const int c = t + 256 * s;
            if (c < NCH) *reinterpret_cast<i32x4*>(ldsc + (c >> 4) * CPS + (c & 15) * 16) = P[s];
        }
    };
    const float4* wl = p.wf + (size_t)nt * p.nj * 64 + lane;
    const int jpt = 8 * p.groups;             // 8-deep k-chunks per tap
    float4 wv[9][2];
    auto load_weights = [&](int g) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int c = 0; c < 2; ++c) wv[tap][c] = wl[(size_t)(tap * jpt + 8 * g + 2 * w + c) * 64];
    };

    constexpr int kTapMin = (-1 * CP - 1) * CPS;
    int poff[MT];
#pragma unroll
    for (int r = 0; r < MT; ++r) {
        const int idx = 32 * r + li;
        poff[r] = (((idx >> 3) + 1) * CP + ((idx & 7) + 1)) * CPS + (16 * w + 4 * lh) * 4 + kTapMin;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    load_weights(0);
    load_patch(0);
#pragma unroll 1
    for (int g = 0; g < p.groups; ++g) {
        if (g > 0) __syncthreads();           // every wave is done reading the previous patch
        store_patch();
        __syncthreads();
        float4 wc[9][2];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) { wc[tap][0] = wv[tap][0]; wc[tap][1] = wv[tap][1]; }
        if (g + 1 < p.groups) { load_weights(g + 1); load_patch(g + 1); }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = ((tap / 3 - 1) * CP + (tap % 3 - 1)) * CPS - kTapMin;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float4 a = wc[tap][c];
#pragma unroll
                for (int r = 0; r < MT; ++r) {
                    const float4 b = *reinterpret_cast<const float4*>(ldsc + poff[r] + toff + c * 32);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[r], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();

    float4* red = reinterpret_cast<float4*>(ldsc);
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            red[((w * MT + r) * 4 + q) * 64 + lane] = make_float4(acc[r][4 * q], acc[r][4 * q + 1], acc[r][4 * q + 2], acc[r][4 * q + 3]);
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d2 = __builtin_amdgcn_make_buffer_rsrc(p.dst2 ? p.dst2 : p.dst, 0, p.dst2 ? p.dst2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.src), 0, p.res ? p.res_bytes : 0u, 0x00020000);
    // bias and residual of every unit first (in flight together): inside the unit loop, behind the stores they may alias, each is its
    // own round trip
    float4 bq[MT], rq[MT];
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        const int unit = t + 256 * u;
        const int ul = unit & 63, q = (unit >> 6) & 3, r = unit >> 8;
        const int idx = 32 * r + (ul & 31);
        const int y = ty0 + (idx >> 3), x = tx0 + (idx & 7);
        const int co = 32 * nt + 8 * q + 4 * (ul >> 5);
        const bool ok = y < p.H && x < p.W && co < p.Cout;
        const unsigned pix = (unsigned)(y * p.W + x);
        bq[u] = (p.bias && ok) ? *reinterpret_cast<const float4*>(p.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        rq[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.res) {
            const i32x4 rr = __builtin_amdgcn_raw_buffer_load_b128(rs_r, ok ? (int)((pix * (unsigned)p.ld_res + (unsigned)co) * 4u) : (int)kOobC, 0, 0);
            rq[u] = make_float4(__int_as_float(rr.x), __int_as_float(rr.y), __int_as_float(rr.z), __int_as_float(rr.w));
        }
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
        const int idx = 32 * r + (ul & 31);
        const int y = ty0 + (idx >> 3), x = tx0 + (idx & 7);
        const int co = 32 * nt + 8 * q + 4 * (ul >> 5);
        const bool ok = y < p.H && x < p.W && co < p.Cout;            // Cout % 4 == 0
        const unsigned pix = (unsigned)(y * p.W + x);
        {
            const float4 b = bq[u];
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
        if (p.act == CIAOSR_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (p.res) { v.x += rq[u].x; v.y += rq[u].y; v.z += rq[u].z; v.w += rq[u].w; }
        i32x4 iv;
        iv.x = __float_as_int(v.x); iv.y = __float_as_int(v.y); iv.z = __float_as_int(v.z); iv.w = __float_as_int(v.w);
        __builtin_amdgcn_raw_buffer_store_b128(iv, rs_d, ok ? (int)((pix * (unsigned)p.ld_dst + (unsigned)co) * 4u) : (int)kOobC, 0, 0);
        if (p.dst2) __builtin_amdgcn_raw_buffer_store_b128(iv, rs_d2, ok ? (int)((pix * (unsigned)p.ld_dst2 + (unsigned)co) * 4u) : (int)kOobC, 0, 0);
    }
}

bool conv3x3_small_ok(int H, int W, int Cin, int Cout, int ld_src, int act) {
    return (long)H * W <= 18432 && Cin >= 64 && (Cin & 63) == 0 && (Cout & 3) == 0 && (ld_src & 3) == 0 &&
           (act == CIAOSR_ACT_NONE || act == CIAOSR_ACT_RELU);
}

int conv3x3_small(const float* src, int ld_src, int H, int W, int Cin, const float* frag, const float* bias, int Cout, float* dst,
                  int ld_dst, float* dst2, int ld_dst2, const float* res, int ld_res, int act, float alpha, hipStream_t s,
                  const char* tag) {
    CIAOSR_CHECK_ARG(src && frag && dst && conv3x3_small_ok(H, W, Cin, Cout, ld_src, act));
    CIAOSR_CHECK_ARG((ld_dst & 3) == 0 && (ld_dst2 & 3) == 0 && (ld_res & 3) == 0 && aligned16(src) && aligned16(frag) && aligned16(dst));
    const size_t M = (size_t)H * W;
    ConvSmallP p;
    p.src = src; p.ld_src = ld_src; p.src_bytes = (unsigned)(((M - 1) * ld_src + Cin) * 4);
    p.H = H; p.W = W; p.tiles_x = ceil_div(W, CT); p.n32 = ceil_div(Cout, 32); p.groups = Cin / 64;
    p.wf = reinterpret_cast<const float4*>(frag); p.nj = 9 * Cin / 8;
    p.bias = bias; p.Cout = Cout;
    p.dst = dst; p.ld_dst = ld_dst; p.dst_bytes = (unsigned)(((M - 1) * ld_dst + Cout) * 4);
    p.dst2 = dst2; p.ld_dst2 = ld_dst2; p.dst2_bytes = dst2 ? (unsigned)(((M - 1) * ld_dst2 + Cout) * 4) : 0u;
    p.res = res; p.ld_res = ld_res; p.res_bytes = res ? (unsigned)(((M - 1) * ld_res + Cout) * 4) : 0u;
    p.act = act; p.alpha = alpha;
    // 8x8-pixel workgroups, or 8x4 when that takes fewer rounds of 256 CUs (see dense_scatter_f32.hip)
    const int wg2 = ceil_div(H, 8) * p.tiles_x * p.n32, wg1 = ceil_div(H, 4) * p.tiles_x * p.n32;
    ProfScope prof(tag ? tag : "conv3x3_small", s);
    if (ceil_div(wg1, 256) < 2 * ceil_div(wg2, 256))
        hipLaunchKernelGGL(conv3x3_small_kernel<1>, dim3(wg1), dim3(256), kConvSmallLds, s, p);
    else
        hipLaunchKernelGGL(conv3x3_small_kernel<2>, dim3(wg2), dim3(256), kConvSmallLds, s, p);
    return launch_status("conv3x3_small");
}

}  // namespace ciaosr
