// Scatter-form dense-block step for SMALL maps (a 48x48 tile): exact-fp32 MFMA, halo-resident, no barrier in the
// main loop.
//
// Step s of a residual dense block (see conv_f32.hip "scatter form"): input group s (64 channels) is convolved once
// with the stacked weight slices of all later dense layers, N = 64 (L - s), K = 576, and accumulated into running sums;
// the 64 columns that belong to layer s itself are finished (bias, ReLU) and become input group s + 1
// (mmedit RDB.layers[l].conv, called from ciaosr_net.py:330-337).
//
// On a 48x48 map a step has 2304 x N outputs and the 128 steps of the trunk are strictly dependent, so a step is a
// latency problem: the generic halo kernel of conv_f32.hip spends ~55 % of a workgroup's cycles outside the MFMA
// (weight stages through LDS with a barrier each, 8-wave K-slice reduction, epilogue).  This kernel keeps the
// decomposition (workgroup = 8x8 pixels x 32 output channels, 72 .. 576 workgroups per step) and removes the rest:
//   * all 18 weight fragments of a wave (9 taps x its 16-channel K slice, pre-packed ciaosr_pack_fragments_f32
//     order) are requested straight from L2 into registers at kernel entry, before the halo patch has even landed;
//   * the 10x10-pixel halo patch (27 KB) is the only LDS staging: one barrier, then 144 MFMAs per wave with one
//     ds_read_b128 per 4 MFMAs and nothing else in between;
//   * 4 waves (K-sliced) instead of 8 and 32 KB of LDS: several workgroups share a CU and overlap each other's
//     prologue / reduction / epilogue with MFMA work.
// Fixed-order K-slice reduction through LDS: deterministic.  Out-of-image accesses go through buffer descriptors.

#include "ops.h"

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int ST = 8;                        // output tile edge (pixels): 64 pixels = 2 MFMA row tiles
constexpr int SP = ST + 2;                   // patch edge with the 1-pixel halo
constexpr int SPS = 272;                     // bytes per patch pixel: 64 fp32 + 16 B pad (conflict-free ds_read_b128)
constexpr int SPATCH = SP * SP * SPS;        // 27 200 B
static_assert(SPATCH <= 32768, "the halo patch must fit the reduction scratch");
constexpr size_t kScatterLds = 32768;        // max(patch, K-slice reduction scratch 4 x 2 x 4 x 64 x 16 B)
constexpr unsigned kOobS = 0xFFFFFFF0u;

#ifdef CIAOSR_PROBE      // developer probe (tools/scatter_probe.hip): cycle stamps of workgroup phases
__device__ unsigned long long g_sprobe[4096 * 8];
#define SPROBE(slot) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_sprobe[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define SPROBE(slot) do { } while (0)
#endif

struct ScatterP {
    float* x; int ldx;                       // block feature buffer [HW][ldx]: group s read, group s + 1 written
    unsigned x_bytes;
    int H, W, tiles_x, n32;                  // n32 = 2 (L - s) output tiles of 32 channels
    int step;
    const float4* wf;                        // fragments [n32][72][64 lanes] float4
    float* acc; int ld_acc; unsigned acc_bytes;   // running sums [HW][64 L]
    const float* bias;                       // [L][64]
};

template <int MT>      // 32-pixel MFMA row tiles per workgroup: 2 (8x8 pixels) or 1 (8 wide x 4 high)
__global__ __launch_bounds__(256) void dense_scatter_small_kernel(ScatterP p) {
    constexpr int TY = 4 * MT, PY = TY + 2;             // tile / patch height
    constexpr int NCH = PY * SP * 16, NLD = (NCH + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldss[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
    const int tile = blockIdx.x / p.n32, nt = blockIdx.x - tile * p.n32;
    const int ty0 = (tile / p.tiles_x) * TY, tx0 = (tile % p.tiles_x) * ST;
    SPROBE(0);

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.x, 0, p.x_bytes, 0x00020000);
    // (1) weights first: 9 taps x 2 chunks of this wave's K slice, k-chunk j = tap*8 + 2w + c
    const float4* wl = p.wf + (size_t)nt * 72 * 64 + lane;
    float4 wv[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < 2; ++c) wv[tap][c] = wl[(tap * 8 + 2 * w + c) * 64];

    // (1b) the running sums this workgroup will add to (independent of the MFMA work: requested now, used in the epilogue)
    const int layer = p.step + (nt >> 1), half = nt & 1;
    const __amdgpu_buffer_rsrc_t rs_acc = __builtin_amdgcn_make_buffer_rsrc(p.acc, 0, p.acc_bytes, 0x00020000);
    float4 prev[MT];
    unsigned aoff[MT], xoff[MT];
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        const int unit = t + 256 * u;
        const int ul = unit & 63, q = (unit >> 6) & 3, r = unit >> 8;
        const int idx = 32 * r + (ul & 31);
        const int y = ty0 + (idx >> 3), x = tx0 + (idx & 7);
        const int col = 32 * half + 8 * q + 4 * (ul >> 5);
        const bool ok = y < p.H && x < p.W;
        const unsigned pix = (unsigned)(y * p.W + x);
        aoff[u] = ok ? (pix * (unsigned)p.ld_acc + (unsigned)(64 * layer + col)) * 4u : kOobS;
        xoff[u] = ok ? (pix * (unsigned)p.ldx + (unsigned)(64 * (p.step + 1) + col)) * 4u : kOobS;
        const i32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(rs_acc, p.step > 0 ? (int)aoff[u] : (int)kOobS, 0, 0);
        prev[u] = make_float4(__int_as_float(pv.x), __int_as_float(pv.y), __int_as_float(pv.z), __int_as_float(pv.w));
    }

    // (2) halo patch of input group `step` -> LDS
    SPROBE(1);
    i32x4 pv_[NLD];
#pragma unroll
    for (int s = 0; s < NLD; ++s) {
        const int c = t + 256 * s;
        const int px = c >> 4, part = c & 15;
        const int py = px / SP, pxx = px - py * SP;
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + pxx;
        const bool ok = c < NCH && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        pv_[s] = __builtin_amdgcn_raw_buffer_load_b128(
            rs, ok ? (int)((unsigned)(gy * p.W + gx) * (unsigned)p.ldx * 4u + (unsigned)(64 * p.step) * 4u + (unsigned)part * 16u) : (int)kOobS, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < NLD; ++s) {
        const int c = t + 256 * s;
        if (c < NCH) *reinterpret_cast<i32x4*>(ldss + (c >> 4) * SPS + (c & 15) * 16) = pv_[s];
    }

    constexpr int kTapMin = (-1 * SP - 1) * SPS;
    int poff[MT];
#pragma unroll
    for (int r = 0; r < MT; ++r) {
        const int idx = 32 * r + li;
        const int y = idx >> 3, x = idx & 7;
        poff[r] = ((y + 1) * SP + (x + 1)) * SPS + (16 * w + 4 * lh) * 4 + kTapMin;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
    __syncthreads();
    SPROBE(2);

    // (3) 144 MFMAs per wave, no barrier
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int toff = ((tap / 3 - 1) * SP + (tap % 3 - 1)) * SPS - kTapMin;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4 a = wv[tap][c];
#pragma unroll
            for (int r = 0; r < MT; ++r) {
                const float4 b = *reinterpret_cast<const float4*>(ldss + poff[r] + toff + c * 32);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[r], 0, 0, 0);
            }
        }
    }
    SPROBE(3);
    __syncthreads();                                       // every wave is done reading the patch

    // (4) K-slice reduction through LDS: red[w][r][q][lane] = accumulator registers 4q..4q+3
    float4* red = reinterpret_cast<float4*>(ldss);
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            red[((w * MT + r) * 4 + q) * 64 + lane] = make_float4(acc[r][4 * q], acc[r][4 * q + 1], acc[r][4 * q + 2], acc[r][4 * q + 3]);
    __syncthreads();

    SPROBE(4);
    // (5) epilogue: 512 float4 units (r, q, lane) = pixel 32r + (lane & 31), channels 8q + 4 (lane >> 5) .. + 3
    float4 v[MT];
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        const int unit = t + 256 * u;
        const int ul = unit & 63, q = (unit >> 6) & 3, r = unit >> 8;
        v[u] = red[((0 * MT + r) * 4 + q) * 64 + ul];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const float4 o = red[((ww * MT + r) * 4 + q) * 64 + ul];
            v[u].x += o.x; v[u].y += o.y; v[u].z += o.z; v[u].w += o.w;
        }
    }
    // the units of a thread share their channel quad: one bias load in front of the stores it may alias
    const float4 bias4 = layer == p.step ? *reinterpret_cast<const float4*>(p.bias + 64 * layer + 32 * half + 8 * ((t >> 6) & 3) + 4 * ((t & 63) >> 5))
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        float4 o = make_float4(v[u].x + prev[u].x, v[u].y + prev[u].y, v[u].z + prev[u].z, v[u].w + prev[u].w);
        i32x4 iv;
        if (layer == p.step) {                             // this layer is complete: bias, ReLU -> input group s + 1
            const float4 b = bias4;
            o.x = fmaxf(o.x + b.x, 0.f); o.y = fmaxf(o.y + b.y, 0.f); o.z = fmaxf(o.z + b.z, 0.f); o.w = fmaxf(o.w + b.w, 0.f);
            iv.x = __float_as_int(o.x); iv.y = __float_as_int(o.y); iv.z = __float_as_int(o.z); iv.w = __float_as_int(o.w);
            __builtin_amdgcn_raw_buffer_store_b128(iv, rs, (int)xoff[u], 0, 0);
        } else {
            iv.x = __float_as_int(o.x); iv.y = __float_as_int(o.y); iv.z = __float_as_int(o.z); iv.w = __float_as_int(o.w);
            __builtin_amdgcn_raw_buffer_store_b128(iv, rs_acc, (int)aoff[u], 0, 0);
        }
    }
    SPROBE(5);
#ifdef CIAOSR_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_sprobe[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg(63492);    // HW_REG_HW_ID
        g_sprobe[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg(63508);    // HW_REG_XCC_ID
    }
#endif
}

int dense_scatter_small(float* X, int ldx, int H, int W, int step, int num_layers, const float* frag, const float* bias_all,
                        float* acc_buf, hipStream_t s) {
    CIAOSR_CHECK_ARG(X && frag && bias_all && acc_buf && (ldx & 3) == 0 && step >= 0 && step < num_layers);
    const size_t xb = (size_t)H * W * ldx * 4, ab = (size_t)H * W * 64 * num_layers * 4;
    CIAOSR_CHECK_ARG(xb < 0xFFFFFF00ull && ab < 0xFFFFFF00ull);
    ScatterP p;
    p.x = X; p.ldx = ldx; p.x_bytes = (unsigned)xb;
    p.H = H; p.W = W; p.tiles_x = ceil_div(W, ST);
    p.n32 = 2 * (num_layers - step);
    p.step = step;
    p.wf = reinterpret_cast<const float4*>(frag);
    p.acc = acc_buf; p.ld_acc = 64 * num_layers; p.acc_bytes = (unsigned)ab;
    p.bias = bias_all;
    // 8x8-pixel workgroups, or 8x4 when that takes fewer rounds of 256 CUs (workgroups of a CU run their MFMA phases back to
    // back: a step costs ceil(WGs / 256) phases; half-size phases quantise finer at twice the weight stream per MAC)
    const int wg2 = ceil_div(H, 8) * p.tiles_x * p.n32, wg1 = ceil_div(H, 4) * p.tiles_x * p.n32;
    const int cost2 = 2 * ceil_div(wg2, 256), cost1 = ceil_div(wg1, 256);       // in half-phase units
    const bool use1 = cost1 < cost2;
    ProfScope prof("enc_dense_scatter", s);
    if (use1)
        hipLaunchKernelGGL(dense_scatter_small_kernel<1>, dim3(wg1), dim3(256), kScatterLds, s, p);
    else
        hipLaunchKernelGGL(dense_scatter_small_kernel<2>, dim3(wg2), dim3(256), kScatterLds, s, p);
    return launch_status("dense_scatter_small");
}

}  // namespace ciaosr
