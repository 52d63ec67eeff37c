// Fused head kernels for the IEEE-half modes with ONE wide workgroup per CU (round 4).  Three instantiations of one template:
//
//   mode        rows / workgroup   weights     activations   MFMAs per product   Z between the two kernels
//   f16         256                half        half          1                   half
//   f16-pairs   256                hi + lo     half          2                   half
//   f16x3       128                hi + lo     hi + lo       3                   fp32        (the fp32-tolerance fast mode)
//
// Why wide.  v_mfma_f32_32x32x16_f16 consumes a 1-KB weight fragment per 32 cycles and SIMD; the weights are the operand every
// workgroup streams in full out of L2, so a CU draws 128 / m B/clk of them at the full MFMA rate when a fragment is used for m row
// tiles ("operand budget").  The round-2/3 kernels (head_fused_h16.hip: 128 rows, two workgroups per CU) run m = 4 and
// sit on that budget: L2 delivers ~14.6 B/clk per CU here, 32 are asked, MFMA-busy 0.46 measured.  More row tiles per fragment need
// the activations of more rows in LDS: [256][264] half = 132 KB = ONE workgroup per CU, 8 waves (two per SIMD), each owning ONE
// 32-column tile of a 256-wide layer for all 8 row tiles (128 accumulator registers): m = 8, 16 B/clk.  With one workgroup per CU
// nothing hides its MFMA-free phases (index math, table gathers, logits) -- the price of the wider tile; measured against the old
// kernels in DESIGN 4.3d.
//
// f16x3 spends the same 132 KB on TWO activation arrays of 128 rows: an activation a is kept as hi = half(a), lo = half(a - hi)
// (~22 mantissa bits) and a product enters the fp32 accumulator as  w_hi a_hi + w_lo a_hi + w_hi a_lo  (the dropped w_lo a_lo is
// ~2^-22 relative).  The f16-pairs mode lands 4 % outside |delta| <= 1e-3 on the full C3 tile because of the 11-bit ACTIVATIONS of
// the three MLP chains (tools/pairs_probe.py); this mode measures max |delta| 2.7e-5, rms 2.1e-6 there.  Half subnormals (a lo half is
// subnormal for |a| < 0.125) pass the matrix pipe unflushed (tools/ubench/f16_denorm.hip).  Per k-step a wave loads w_hi and w_lo
// (2 KB) for 12 MFMAs: 21 B/clk per CU at the full rate, i.e. the L2 stream caps it at ~0.70 -- the probe shows the hidden layers
// exactly there (17.4k cycles per layer against 12.3k of MFMA issue).
//
// Same row order as head_fused_h16.hip generalised to QW = rows / 4 queries: row m = QW j + q (j = key sample), so the four samples of
// a query are four row tiles of ONE lane and z = sum_j a_j value_j . w_v,j is four FMAs on the lane's own accumulators; swapped MFMA
// operands (weights = A from L2 in pre-packed fragment order, activations = B from LDS via ds_read_b128); hidden layers in place.
// Round 6: compiled per element type like the other 16-bit units (-DCIAOSR_F16=1 -> ciaosr::f16::wide, =0 -> ciaosr::b16::wide).  The bf16
// build carries the pair mode only ("bf16x3": bf16 hi + lo weights AND activations, 16 mantissa bits each, three MFMAs per product, fp32 Z):
// what the SwinIR-CiaoSR head (BASELINE config 5, "bf16") runs, whose 8-bit single-bf16 activations miss the 0.01 dB gate (0.060 dB).
#include "h16_util.h"
#include "index_math.h"
#include "ops.h"

namespace ciaosr {
namespace CIAOSR_H16_NS {
namespace wide {

constexpr bool kF16 = CIAOSR_F16 != 0;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

#ifdef CIAOSR_PROBE      // developer probe build (make probe; tools/head_probe.py 192 f16x3 | f16w): cycle stamps of workgroup phases
__device__ unsigned long long g_xprobe[4096 * 16];
#define XPROBE(slot) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_xprobe[blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define XPROBE(slot) do { } while (0)
#endif

constexpr int XH = 256;            // hidden width
constexpr int XLD = XH + 8;        // LDS row stride in halves (528 B: conflict-free ds_read_b128)
constexpr int XKS = XH / 16;       // k-steps per 256-wide layer
constexpr int XNW = 8;             // waves per workgroup
constexpr int XNT = XNW * 64;      // threads
constexpr unsigned kOobX = 0xFFFFFFF0u;

// ROWS activation rows per workgroup; WP: weights as hi + lo pairs; AP: activations as hi + lo pairs (two LDS arrays, fp32 Z)
template <int ROWS_, bool WP_, bool AP_>
struct Mode {
    static constexpr int ROWS = ROWS_;
    static constexpr bool WP = WP_, AP = AP_;
    static constexpr int MI = ROWS / 32;          // 32-row MFMA tiles
    static constexpr int QW = ROWS / 4;           // queries per kv workgroup
    static constexpr int H2 = QW / 32;            // 32-query halves: tile mi = H2 j + h
    static constexpr size_t ACT = (size_t)ROWS * XLD * 2;                   // one activation array
    static constexpr size_t ACTS = ACT * (AP ? 2 : 1);
    static constexpr size_t KV_LDS = ACTS + (size_t)(ROWS * 4 + XNW * ROWS + ROWS) * sizeof(float) + (ROWS + QW + ROWS) * sizeof(int);
};
using ModeF16 = Mode<256, false, false>;
using ModePairs = Mode<256, true, false>;
using ModeX3 = Mode<128, true, true>;

__device__ __forceinline__ float4 xload4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ void xstore4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, float4 v) {
    i32x4 iv;
    iv.x = __float_as_int(v.x); iv.y = __float_as_int(v.y); iv.z = __float_as_int(v.z); iv.w = __float_as_int(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(iv, rsrc, (int)byte_off, 0, 0);
}

// (a, b) -> packed halves hi = half(clamp(x)), lo = half(clamp(x) - hi); RELU: clamp = [0, 65504], else [-65504, 65504]
template <bool RELU>
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    if constexpr (!kF16) {      // bf16: no saturation (fp32 range); hi = bf16(x) to nearest even, lo = bf16(x - hi) (the residual is exact in fp32)
        const float va = RELU ? fmaxf(a, 0.f) : a, vb = RELU ? fmaxf(b, 0.f) : b;
        hi = pack_h16x2<false>(va, vb);
        lo = pack_h16x2<false>(va - h16_lo<false>(hi), vb - h16_hi<false>(hi));
        return;
    }
    const float floor_ = RELU ? 0.f : -kHalfMax;
    const f32x2_t v = {__builtin_amdgcn_fmed3f(a, floor_, kHalfMax), __builtin_amdgcn_fmed3f(b, floor_, kHalfMax)};
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);
    // residual x - hi (exact in fp32) as ONE v_fma_mix_f32 per element (fma of the half operand, taken straight from its half of the
    // packed register, with -1.0 and the fp32 x) instead of a convert and a subtract (hipcc folds the fma(x, -1, y) form back into
    // cvt + sub, hence the asm)
    const unsigned hp = __builtin_bit_cast(unsigned, h);
    f32x2_t r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(hp), "v"(v.x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(hp), "v"(v.y));
    hi = hp;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}
// hi + lo of packed element 0 / 1 back to fp32 (one v_fma_mix_f32 each: hi * 1.0 + lo, both half operands from their packed registers)
__device__ __forceinline__ float pair0(unsigned hi, unsigned lo) {
    if constexpr (!kF16) return h16_lo<false>(hi) + h16_lo<false>(lo);
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi), "v"(lo));
    return r;
}
__device__ __forceinline__ float pair1(unsigned hi, unsigned lo) {
    if constexpr (!kF16) return h16_hi<false>(hi) + h16_hi<false>(lo);
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi), "v"(lo));
    return r;
}
// element 0 / 1 of a packed activation pair (AP: hi + lo) as fp32
template <bool AP>
__device__ __forceinline__ float act0(unsigned hi, unsigned lo) { if constexpr (AP) return pair0(hi, lo); else return h16_lo<kF16>(hi); }
template <bool AP>
__device__ __forceinline__ float act1(unsigned hi, unsigned lo) { if constexpr (AP) return pair1(hi, lo); else return h16_hi<kF16>(hi); }

__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) { return mfma_h16<kF16>(a, b, c); }

// Weight ring: RD stages, requested RD - 1 k-steps ahead.  A k-step of the 256-row modes is only 8 MFMAs = 256 cycles of the SIMD's
// pipe, shared by its two waves: two steps ahead (the 128-row kernels' depth) is ~1000 cycles at best, about one L2 round trip under
// load -- the probe showed the hidden layers at 68 % with the weight stream HALVED, i.e. latency-, not bandwidth-bound.
template <bool WP, bool AP> constexpr int ring_depth() { return AP ? 3 : (WP ? 4 : 6); }

// first RD - 1 weight stages of a pass: requested early so that they are in flight across the barrier in front of the pass
template <bool WP, int RD>
__device__ __forceinline__ void load_w01(const uint4* __restrict__ wh, const uint4* __restrict__ wl, uint4 (&fh)[RD], uint4 (&fl)[RD]) {
#pragma unroll
    for (int i = 0; i + 1 < RD; ++i) {
        fh[i] = wh[64 * i];
        if constexpr (WP) fl[i] = wl[64 * i];
    }
}

// acc[i] += W . X^T over NKS (>= 2) k-steps of one 32-column weight tile, for the NM row tiles at rows 32 * tile(i) of the
// activation array(s); xh / xl: &X[lane row][8 g] of row tile 0, TS = halves between consecutive row tiles of the pass.
// WP adds the w_lo . x_hi products, AP the w_hi . x_lo ones (small terms first, each product type over the independent accumulators).
template <int NKS, int NM, int TS, bool WP, bool AP, int RD>
__device__ __forceinline__ void mma_pass(const unsigned short* xh, const unsigned short* xl, const uint4* __restrict__ wh,
                                         const uint4* __restrict__ wl, f32x16 (&acc)[NM], uint4 (&fh)[RD], uint4 (&fl)[RD]) {
    uint4 ah[2][NM], al[2][AP ? NM : 1];
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        ah[0][i] = *reinterpret_cast<const uint4*>(xh + i * TS);
        if constexpr (AP) al[0][i] = *reinterpret_cast<const uint4*>(xl + i * TS);
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + RD - 1 < NKS) {
            fh[(ks + RD - 1) % RD] = wh[(long)(ks + RD - 1) * 64];
            if constexpr (WP) fl[(ks + RD - 1) % RD] = wl[(long)(ks + RD - 1) * 64];
        }
        if (ks + 1 < NKS) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                ah[(ks + 1) & 1][i] = *reinterpret_cast<const uint4*>(xh + i * TS + 16 * (ks + 1));
                if constexpr (AP) al[(ks + 1) & 1][i] = *reinterpret_cast<const uint4*>(xl + i * TS + 16 * (ks + 1));
            }
        }
        const uint4 w_hi = fh[ks % RD];
        if constexpr (WP) {
            const uint4 w_lo = fl[ks % RD];
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i] = mfma(w_lo, ah[ks & 1][i], acc[i]);
        }
        if constexpr (AP) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i] = mfma(w_hi, al[ks & 1][i], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i] = mfma(w_hi, ah[ks & 1][i], acc[i]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// accumulators of the wave's 32-column tile initialised with the bias (channel 8 g + 4 lh + e of the tile <-> acc[i][4 g + e])
template <int NM>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NM], const float* __restrict__ bias32, int lh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(bias32 + 8 * g + 4 * lh);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            acc[i][4 * g] = b.x; acc[i][4 * g + 1] = b.y; acc[i][4 * g + 2] = b.z; acc[i][4 * g + 3] = b.w;
        }
    }
}

// four fp32 activations -> LDS: relu (saturating), AP: split into the two arrays
template <bool AP, bool RELU>
__device__ __forceinline__ void store_act4(unsigned short* Xh, unsigned short* Xl, int off, float a, float b, float c, float d) {
    if constexpr (AP) {
        uint2 h, l;
        split2<RELU>(a, b, h.x, l.x);
        split2<RELU>(c, d, h.y, l.y);
        *reinterpret_cast<uint2*>(Xh + off) = h;
        *reinterpret_cast<uint2*>(Xl + off) = l;
    } else {
        uint2 o;
        o.x = RELU ? pack_relu_h16x2<kF16>(a, b) : pack_h16x2<kF16>(a, b);
        o.y = RELU ? pack_relu_h16x2<kF16>(c, d) : pack_h16x2<kF16>(c, d);
        *reinterpret_cast<uint2*>(Xh + off) = o;
    }
}

// relu + store of the wave's accumulators, columns [col0, col0 + 32)
template <typename M>
__device__ __forceinline__ void store_relu(unsigned short* Xh, unsigned short* Xl, const f32x16 (&acc)[M::MI], int col0, int li, int lh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = col0 + 8 * g + 4 * lh;
#pragma unroll
        for (int mi = 0; mi < M::MI; ++mi)
            store_act4<M::AP, true>(Xh, Xl, (32 * mi + li) * XLD + col, acc[mi][4 * g], acc[mi][4 * g + 1], acc[mi][4 * g + 2], acc[mi][4 * g + 3]);
    }
}

// hidden layer, in place.  Entry: X may still be being written by other waves (the leading barrier orders it); exit: X holds this
// layer's activations as far as THIS wave's stores go -- the next consumer starts with a barrier.
template <typename M>
__device__ __forceinline__ void hidden_layer(unsigned short* Xh, unsigned short* Xl, const void* __restrict__ frag, const void* __restrict__ frag_lo,
                                             const float* __restrict__ bias, int w, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const uint4* wh = reinterpret_cast<const uint4*>(frag) + (size_t)w * XKS * 64 + lane;
    const uint4* wl = reinterpret_cast<const uint4*>(M::WP ? frag_lo : frag) + (size_t)w * XKS * 64 + lane;
    constexpr int RD = ring_depth<M::WP, M::AP>();
    uint4 fh[RD], fl[RD];
    load_w01<M::WP, RD>(wh, wl, fh, fl);
    f32x16 acc[M::MI];
    init_bias<M::MI>(acc, bias + 32 * w, lh);
    __syncthreads();
    mma_pass<XKS, M::MI, 32 * XLD, M::WP, M::AP, RD>(Xh + li * XLD + 8 * lh, Xl + li * XLD + 8 * lh, wh, wl, acc, fh, fl);
    __syncthreads();
    store_relu<M>(Xh, Xl, acc, 32 * w, li, lh);
}

// layer-0 rows from the hoisted fp32 tables: ROWS / 8 rows per thread, the table gathers 16 at a time in flight (no accumulator is live)
template <typename M>
__device__ __forceinline__ void build_rows(unsigned short* Xh, unsigned short* Xl, const FusedChain& c, const int* s_kpix, const float* s_t4, int t) {
    const int n4 = t & 63, r0 = t >> 6;
    float4 tw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) tw[e] = *reinterpret_cast<const float4*>(c.tail + (size_t)(4 * n4 + e) * c.ld_tail);
    constexpr int NR = M::ROWS / XNW;
#pragma unroll
    for (int b = 0; b < NR; b += 16) {
        float4 tv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) tv[i] = reinterpret_cast<const float4*>(c.table + (size_t)s_kpix[r0 + XNW * (b + i)] * XH)[n4];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + XNW * (b + i);
            const float4 q = *reinterpret_cast<const float4*>(s_t4 + 4 * r);       // rel_y rel_x scale_y scale_x
            store_act4<M::AP, true>(Xh, Xl, r * XLD + 4 * n4,
                                    tv[i].x + tw[0].x * q.x + tw[0].y * q.y + tw[0].z * q.z + tw[0].w * q.w,
                                    tv[i].y + tw[1].x * q.x + tw[1].y * q.y + tw[1].z * q.z + tw[1].w * q.w,
                                    tv[i].z + tw[2].x * q.x + tw[2].y * q.y + tw[2].z * q.z + tw[2].w * q.w,
                                    tv[i].w + tw[3].x * q.x + tw[3].y * q.y + tw[3].z * q.z + tw[3].w * q.w);
        }
    }
}

// z (four channels of one query) -> Z: fp32 for the activation-pair mode, half otherwise
template <bool AP>
__device__ __forceinline__ void store_z(__amdgpu_buffer_rsrc_t rs_z, unsigned zoff, int d0, bool ok, float4 z) {
    if constexpr (AP) {
        xstore4(rs_z, ok ? zoff + (unsigned)d0 * 4u : kOobX, z);
    } else {
        const uint2 zb = pack_h16x4<kF16>(z.x, z.y, z.z, z.w);
        i32x2 zi; zi.x = (int)zb.x; zi.y = (int)zb.y;
        __builtin_amdgcn_raw_buffer_store_b64(zi, rs_z, (int)(ok ? zoff + (unsigned)d0 * 2u : kOobX), 0, 0);
    }
}

template <typename M>
__global__ __launch_bounds__(XNT) void head_kv_fused_wide_kernel(FusedKVP p) {
    constexpr int ROWS = M::ROWS, MI = M::MI, QW = M::QW, H2 = M::H2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* Xh = reinterpret_cast<unsigned short*>(smem_raw);                     // [ROWS][264] half (hi)
    unsigned short* Xl = reinterpret_cast<unsigned short*>(smem_raw + (M::AP ? M::ACT : 0));   // AP: [ROWS][264] half (lo)
    float* s_t4 = reinterpret_cast<float*>(smem_raw + M::ACTS);                           // [ROWS][4]
    float* s_part = s_t4 + ROWS * 4;                                                      // [8][ROWS]
    float* s_attn = s_part + XNW * ROWS;                                                  // [ROWS]
    int* s_kpix = reinterpret_cast<int*>(s_attn + ROWS);                                  // [ROWS]
    int* s_qpix = s_kpix + ROWS;                                                          // [QW]
    int* s_goff = s_qpix + QW;                                                            // [ROWS]

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qbase = blockIdx.x * QW;

    XPROBE(0);
    // ---- index math: row m = QW j + q -----------------------------------------------------------------------------
    int bad = 0;
    if (t < ROWS) {
        const int qi = t % QW, j = t / QW;
        const int ql = qbase + qi;
        int kpix = 0, goff = -1;
        float t4[4] = {0.f, 0.f, 0.f, 0.f};
        if (ql < p.nq) {
            const long q = p.q0 + ql;
            const float cy = p.coord[2 * q], cx = p.coord[2 * q + 1];
            const long c0 = p.chunk > 0 ? (q / p.chunk) * p.chunk : 0;
            const KeySample s = key_sample(cy, cx, p.cell[2 * c0], p.cell[2 * c0 + 1], p.H, p.W, j, 2);
            kpix = s.ky * p.W + s.kx;
            t4[0] = s.rel_y; t4[1] = s.rel_x;
            t4[2] = mul_rn(p.cell[2 * q], (float)p.H);
            t4[3] = mul_rn(p.cell[2 * q + 1], (float)p.W);
            const int iy = nearest_index(cy, p.H), ix = nearest_index(cx, p.W);
            const bool qin = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            if (j == 0) s_qpix[qi] = qin ? iy * p.W + ix : -1;
            if (qin) {
                const int oy = s.ky - iy, ox = s.kx - ix;
                if (oy >= -1 && oy <= 1 && ox >= -1 && ox <= 1) goff = (iy * p.W + ix) * 9 + (oy + 1) * 3 + (ox + 1);
                else bad = 1;
            }
        } else if (j == 0) {
            s_qpix[qi] = -1;
        }
        s_kpix[t] = kpix;
        s_goff[t] = goff;
        *reinterpret_cast<float4*>(s_t4 + 4 * t) = make_float4(t4[0], t4[1], t4[2], t4[3]);
    }
    const bool table = p.G != nullptr && !__syncthreads_or(bad);
    if (p.G == nullptr) __syncthreads();

    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U), 0, p.u_bytes, 0x00020000);

    // ================= phi_k =====================================================================
    XPROBE(1);
    build_rows<M>(Xh, Xl, p.k, s_kpix, s_t4, t);
    XPROBE(2);
    for (int l = 0; l < p.k.n_hidden; ++l) {
        hidden_layer<M>(Xh, Xl, p.k.frag_hidden[l], p.k.frag_hidden_lo[l], p.k.bias_hidden[l], w, lane);
        if (l < 3) XPROBE(10 + l);
    }
    __syncthreads();
    XPROBE(3);
    if (table) {
        // logit = h4 . G[query pixel, key offset] + c (fp32 table): TPR threads per row, 16 gathers in flight
        constexpr int TPR = XNT / ROWS, NB = XH / (4 * TPR * 16);      // x3: 4 threads x 1 batch; 256 rows: 2 threads x 2 batches
        const int row = t / TPR, part = t % TPR;
        const int go = s_goff[row];
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, p.g_bytes, 0x00020000);
        const unsigned gbase = go >= 0 ? (unsigned)go * (unsigned)p.ldg * 4u : kOobX;
        float a = (go >= 0 && part == 0) ? p.G[(size_t)go * p.ldg + 256] : 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float4 gv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) gv[i] = xload4(rs_g, gbase == kOobX ? kOobX : gbase + (unsigned)(4 * TPR * (16 * b + i) + 4 * part) * 4u);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int col = 4 * TPR * (16 * b + i) + 4 * part;
                const uint2 xh = *reinterpret_cast<const uint2*>(Xh + row * XLD + col);
                uint2 xl = xh;
                if constexpr (M::AP) xl = *reinterpret_cast<const uint2*>(Xl + row * XLD + col);
                a += act0<M::AP>(xh.x, xl.x) * gv[i].x + act1<M::AP>(xh.x, xl.x) * gv[i].y + act0<M::AP>(xh.y, xl.y) * gv[i].z +
                     act1<M::AP>(xh.y, xl.y) * gv[i].w;
            }
        }
        a += quad_xor1(a);
        if constexpr (TPR == 4) a += quad_xor2(a);
        // slab `part` gets the logit (part 0) or 0; the remaining slabs are zeroed TPR at a time
        s_part[part * ROWS + row] = part == 0 ? a : 0.f;
#pragma unroll
        for (int k = TPR; k < XNW; k += TPR) s_part[(part + k) * ROWS + row] = 0.f;
    } else {
        // fallback (no table, or a key outside the query's 3x3 neighbourhood): imnet_k's output layer on the MFMA
        float part[MI];
        unsigned koff[MI], qoff[H2];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            part[mi] = 0.f;
            koff[mi] = (unsigned)s_kpix[32 * mi + li] * (unsigned)p.ldu * 4u;
        }
#pragma unroll
        for (int h = 0; h < H2; ++h) {
            const int qp = s_qpix[32 * h + li];
            qoff[h] = qp >= 0 ? (unsigned)qp * (unsigned)p.ldu * 4u : kOobX;
        }
        const int n_units = (p.k.n_out + 31) >> 5;
        const __amdgpu_buffer_rsrc_t rs_bk =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.k.bias_out), 0, (unsigned)p.k.n_out * 4u, 0x00020000);
        for (int u = w; u < n_units; u += XNW) {
            const uint4* wh = reinterpret_cast<const uint4*>(p.k.frag_out) + (size_t)u * XKS * 64 + lane;
            const uint4* wl = reinterpret_cast<const uint4*>(M::WP ? p.k.frag_out_lo : p.k.frag_out) + (size_t)u * XKS * 64 + lane;
            constexpr int RD = ring_depth<M::WP, M::AP>();
            uint4 fh[RD], fl[RD];
            load_w01<M::WP, RD>(wh, wl, fh, fl);
            f32x16 acc[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
            mma_pass<XKS, MI, 32 * XLD, M::WP, M::AP, RD>(Xh + li * XLD + 8 * lh, Xl + li * XLD + 8 * lh, wh, wl, acc, fh, fl);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * u + 8 * g + 4 * lh;
                const unsigned doff = d0 < p.k.n_out ? (unsigned)d0 * 4u : kOobX;
                const float4 bv = xload4(rs_bk, doff);
                float4 qv[H2];
#pragma unroll
                for (int h = 0; h < H2; ++h) qv[h] = xload4(rs_u, (doff == kOobX || qoff[h] == kOobX) ? kOobX : qoff[h] + doff);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const float4 kv = xload4(rs_u, doff == kOobX ? kOobX : koff[mi] + doff);
                    const float4 q4 = qv[mi % H2];
                    part[mi] += q4.x * (kv.x * (acc[mi][4 * g] + bv.x)) + q4.y * (kv.y * (acc[mi][4 * g + 1] + bv.y)) +
                                q4.z * (kv.z * (acc[mi][4 * g + 2] + bv.z)) + q4.w * (kv.w * (acc[mi][4 * g + 3] + bv.w));
                }
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            part[mi] += __shfl_xor(part[mi], 32, 64);
            if (lh == 0) s_part[w * ROWS + 32 * mi + li] = part[mi];
        }
    }
    __syncthreads();
    if (t < QW) {               // 4-way softmax of query t: rows t, t + QW, t + 2 QW, t + 3 QW
        float lg[4], m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = QW * j + t;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < XNW; ++k) sum += s_part[k * ROWS + row];
            lg[j] = sum / p.softmax_scale;
            m = fmaxf(m, lg[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { lg[j] = expf(lg[j] - m); den += lg[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) s_attn[QW * j + t] = lg[j] / den;
    }

    // ================= phi_v =====================================================================
    XPROBE(4);
    build_rows<M>(Xh, Xl, p.v, s_kpix, s_t4, t);          // every wave is past its logit reads of X (barrier above)
    XPROBE(5);
    for (int l = 0; l < p.v.n_hidden; ++l) {
        hidden_layer<M>(Xh, Xl, p.v.frag_hidden[l], p.v.frag_hidden_lo[l], p.v.bias_hidden[l], w, lane);
        if (l < 3) XPROBE(13 + l);
    }
    __syncthreads();
    XPROBE(6);
    {
        // output layer of imnet_v fused with z = sum_j a_j value_j . w_v,j: a 32-column unit per wave and turn.  The four samples of a
        // query are the row tiles H2 j + h of one lane, so a unit runs H2 passes of four row tiles each (h = query half): in the first
        // the unit's hi fragments are streamed from L2 INTO 16 registers each holding a k-step, the second pass reads them there (the
        // lo fragments of the pair modes are streamed again: 3 instead of 4 KB per k-step for the two passes)
        const int n_units = (p.v.n_out + 31) >> 5;
        const __amdgpu_buffer_rsrc_t rs_bv =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.v.bias_out), 0, (unsigned)p.v.n_out * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_z =
            __builtin_amdgcn_make_buffer_rsrc(p.Z, 0, (unsigned)((size_t)p.nq * p.ldz * (M::AP ? 4 : 2)), 0x00020000);
        for (int u = w; u < n_units; u += XNW) {
            const uint4* wh = reinterpret_cast<const uint4*>(p.v.frag_out) + (size_t)u * XKS * 64 + lane;
            const uint4* wl = reinterpret_cast<const uint4*>(M::WP ? p.v.frag_out_lo : p.v.frag_out) + (size_t)u * XKS * 64 + lane;
            uint4 wreg[H2 > 1 ? XKS : 1];
#pragma unroll
            for (int h = 0; h < H2; ++h) {
                unsigned voff[4];
                float av[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    voff[j] = (unsigned)s_kpix[QW * j + 32 * h + li] * (unsigned)p.ldu * 4u;
                    av[j] = s_attn[QW * j + 32 * h + li];
                }
                const int ql = qbase + 32 * h + li;
                const unsigned zoff = ql < p.nq ? (unsigned)ql * (unsigned)p.ldz * (M::AP ? 4u : 2u) : kOobX;
                // the pass's value rows and bias: in flight under its MFMAs
                float4 vv[4][4];
                f32x16 acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 32 * u + 8 * g + 4 * lh;
                    const unsigned doff = d0 < p.v.n_out ? (unsigned)d0 * 4u : kOobX;
                    const float4 bv = xload4(rs_bv, doff);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        vv[g][j] = xload4(rs_u, doff == kOobX ? kOobX : voff[j] + doff);
                        acc[j][4 * g] = bv.x; acc[j][4 * g + 1] = bv.y; acc[j][4 * g + 2] = bv.z; acc[j][4 * g + 3] = bv.w;
                    }
                }
                const unsigned short* xh = Xh + (32 * h + li) * XLD + 8 * lh;
                const unsigned short* xl = Xl + (32 * h + li) * XLD + 8 * lh;
                if constexpr (H2 == 1) {
                    uint4 fh[3], fl[3];
                    load_w01<M::WP, 3>(wh, wl, fh, fl);
                    mma_pass<XKS, 4, QW * XLD, M::WP, M::AP, 3>(xh, xl, wh, wl, acc, fh, fl);
                } else {
                    // (the two-pass form exists for the 256-row modes only: half activations, AP = false)
                    // a k-step is 4 MFMAs = 128 cycles here: the unit's 16 hi fragments are ALL requested up front (they have
                    // their registers anyway), the lo ring runs 5 steps ahead
                    constexpr int RL = 4;
                    uint4 fl[RL];
                    uint4 ah[2][4];
                    if (h == 0) {
#pragma unroll
                        for (int ks = 0; ks < XKS; ++ks) wreg[ks] = wh[(long)ks * 64];
                    }
                    if constexpr (M::WP) {
#pragma unroll
                        for (int i = 0; i + 1 < RL; ++i) fl[i] = wl[64 * i];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) ah[0][j] = *reinterpret_cast<const uint4*>(xh + j * QW * XLD);
#pragma unroll
                    for (int ks = 0; ks < XKS; ++ks) {
                        if (ks + RL - 1 < XKS) {
                            if constexpr (M::WP) fl[(ks + RL - 1) % RL] = wl[(long)(ks + RL - 1) * 64];
                        }
                        if (ks + 1 < XKS) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) ah[(ks + 1) & 1][j] = *reinterpret_cast<const uint4*>(xh + j * QW * XLD + 16 * (ks + 1));
                        }
                        if constexpr (M::WP) {
                            const uint4 w_lo = fl[ks % RL];
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j] = mfma(w_lo, ah[ks & 1][j], acc[j]);
                        }
                        const uint4 w_hi = wreg[ks];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = mfma(w_hi, ah[ks & 1][j], acc[j]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        z.x = fmaf(av[j] * vv[g][j].x, acc[j][4 * g], z.x);
                        z.y = fmaf(av[j] * vv[g][j].y, acc[j][4 * g + 1], z.y);
                        z.z = fmaf(av[j] * vv[g][j].z, acc[j][4 * g + 2], z.z);
                        z.w = fmaf(av[j] * vv[g][j].w, acc[j][4 * g + 3], z.w);
                    }
                    const int d0 = 32 * u + 8 * g + 4 * lh;
                    store_z<M::AP>(rs_z, zoff, d0, zoff != kOobX && d0 < p.v.n_out, z);
                }
            }
        }
    }
    XPROBE(7);
#ifdef CIAOSR_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_xprobe[blockIdx.x * 16 + 8] = __builtin_amdgcn_s_getreg(63492);    // HW_REG_HW_ID
        g_xprobe[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_getreg(63508);    // HW_REG_XCC_ID
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// decode: ROWS queries per workgroup; the input layer is streamed through LDS in 256-column chunks of Z (half rows copied as they
// are; fp32 rows of the activation-pair mode split into hi + lo on the way in).
// One chunk: [weight prefetch] [barrier] [ROWS x 256 piece of Z -> LDS] [barrier] [NKS k-steps].  Columns >= kc are zero in LDS and
// the weight stream is read through a bounded buffer descriptor (0 beyond its end): a ragged last chunk may run whole k-steps past Dv.
template <typename M, int NKS>
__device__ __forceinline__ void decode_chunk(unsigned short* Xh, unsigned short* Xl, const FusedQP& p, __amdgpu_buffer_rsrc_t rs_z,
                                             __amdgpu_buffer_rsrc_t rs_wh, __amdgpu_buffer_rsrc_t rs_wl, int k0, int kc, int qbase, int t, int w,
                                             int lane, f32x16 (&acc)[M::MI]) {
    constexpr int MI = M::MI;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned wbase = ((unsigned)w * (unsigned)p.nj_in + (unsigned)(k0 >> 4)) * 64u * 16u + (unsigned)lane * 16u;
    auto wload = [&](__amdgpu_buffer_rsrc_t rs, int ks) -> uint4 {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(wbase + (unsigned)ks * 1024u), 0, 0);
        return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
    };
    constexpr int RD = ring_depth<M::WP, M::AP>() < NKS ? ring_depth<M::WP, M::AP>() : (NKS > 2 ? NKS : 3);
    uint4 fh[RD], fl[RD];
#pragma unroll
    for (int i = 0; i + 1 < RD; ++i) {
        fh[i] = wload(rs_wh, i);
        if constexpr (M::WP) fl[i] = wload(rs_wl, i);
    }
    if (k0 > 0) __syncthreads();
    if constexpr (M::AP) {
        const int c4 = (t & 63) * 4, r0 = t >> 6;
        float4 zv[M::ROWS / XNW];
#pragma unroll
        for (int i = 0; i < M::ROWS / XNW; ++i) {
            const int ql = qbase + r0 + XNW * i;
            zv[i] = xload4(rs_z, (ql < p.nq && c4 < kc) ? ((unsigned)ql * (unsigned)p.ldz + (unsigned)(k0 + c4)) * 4u : kOobX);
        }
#pragma unroll
        for (int i = 0; i < M::ROWS / XNW; ++i)
            store_act4<true, false>(Xh, Xl, (r0 + XNW * i) * XLD + c4, zv[i].x, zv[i].y, zv[i].z, zv[i].w);
    } else {
        const int c8 = (t & 31) * 8, r0 = t >> 5;           // 16 rows x 32 pieces of 16 B per instruction; ROWS / 16 of them per thread
        constexpr int NL = M::ROWS / 16;
        i32x4 zv[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int ql = qbase + r0 + 16 * i;
            zv[i] = __builtin_amdgcn_raw_buffer_load_b128(
                rs_z, (int)((ql < p.nq && c8 < kc) ? ((unsigned)ql * (unsigned)p.ldz + (unsigned)(k0 + c8)) * 2u : kOobX), 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) *reinterpret_cast<i32x4*>(Xh + (r0 + 16 * i) * XLD + c8) = zv[i];
    }
    __syncthreads();
    const unsigned short* xh = Xh + li * XLD + 8 * lh;
    const unsigned short* xl = Xl + li * XLD + 8 * lh;
    uint4 ah[2][MI], al[2][M::AP ? MI : 1];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        ah[0][mi] = *reinterpret_cast<const uint4*>(xh + mi * 32 * XLD);
        if constexpr (M::AP) al[0][mi] = *reinterpret_cast<const uint4*>(xl + mi * 32 * XLD);
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + RD - 1 < NKS) {
            fh[(ks + RD - 1) % RD] = wload(rs_wh, ks + RD - 1);
            if constexpr (M::WP) fl[(ks + RD - 1) % RD] = wload(rs_wl, ks + RD - 1);
        }
        if (ks + 1 < NKS) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                ah[(ks + 1) & 1][mi] = *reinterpret_cast<const uint4*>(xh + mi * 32 * XLD + 16 * (ks + 1));
                if constexpr (M::AP) al[(ks + 1) & 1][mi] = *reinterpret_cast<const uint4*>(xl + mi * 32 * XLD + 16 * (ks + 1));
            }
        }
        const uint4 w_hi = fh[ks % RD];
        if constexpr (M::WP) {
            const uint4 w_lo = fl[ks % RD];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = mfma(w_lo, ah[ks & 1][mi], acc[mi]);
        }
        if constexpr (M::AP) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = mfma(w_hi, al[ks & 1][mi], acc[mi]);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi] = mfma(w_hi, ah[ks & 1][mi], acc[mi]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// last Linear (256 -> 3) in fp32 on the LDS activations + bilinear/border residual (net:107-108,221): XNT / ROWS threads per row
template <typename M>
__device__ __forceinline__ void decode_tail(const unsigned short* Xh, const unsigned short* Xl, const FusedQP& p, int t, int qbase) {
    constexpr int TPR = XNT / M::ROWS, NC = XH / TPR;
    const int row = t / TPR, part = t % TPR;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    {
        const unsigned short* xh = Xh + row * XLD + NC * part;
        const unsigned short* xl = Xl + row * XLD + NC * part;
        const float* w0 = p.w_last + NC * part;
        const float* w1 = w0 + p.ld_last;
        const float* w2 = w1 + p.ld_last;
#pragma unroll 4
        for (int n = 0; n < NC; n += 4) {
            const uint2 hb = *reinterpret_cast<const uint2*>(xh + n);
            uint2 lb = hb;
            if constexpr (M::AP) lb = *reinterpret_cast<const uint2*>(xl + n);
            const float x0 = act0<M::AP>(hb.x, lb.x), x1 = act1<M::AP>(hb.x, lb.x), x2 = act0<M::AP>(hb.y, lb.y), x3 = act1<M::AP>(hb.y, lb.y);
            const float4 u0 = *reinterpret_cast<const float4*>(w0 + n);
            const float4 u1 = *reinterpret_cast<const float4*>(w1 + n);
            const float4 u2 = *reinterpret_cast<const float4*>(w2 + n);
            a0 += x0 * u0.x + x1 * u0.y + x2 * u0.z + x3 * u0.w;
            a1 += x0 * u1.x + x1 * u1.y + x2 * u1.z + x3 * u1.w;
            a2 += x0 * u2.x + x1 * u2.y + x2 * u2.z + x3 * u2.w;
        }
    }
    a0 += quad_xor1(a0); a1 += quad_xor1(a1); a2 += quad_xor1(a2);
    if constexpr (TPR == 4) { a0 += quad_xor2(a0); a1 += quad_xor2(a1); a2 += quad_xor2(a2); }
    const int ql = qbase + row;
    if (part == 0 && ql < p.nq) {
        const long q = p.q0 + ql;
        float v[3] = {a0 + p.b_last[0], a1 + p.b_last[1], a2 + p.b_last[2]};
        if (p.x_lr) {
            const float cy = p.coord[2 * q], cx = p.coord[2 * q + 1];
            float fy = sub_rn(mul_rn(add_rn(cy, 1.0f), (float)p.H * 0.5f), 0.5f);
            float fx = sub_rn(mul_rn(add_rn(cx, 1.0f), (float)p.W * 0.5f), 0.5f);
            fy = fminf((float)(p.H - 1), fmaxf(fy, 0.f));
            fx = fminf((float)(p.W - 1), fmaxf(fx, 0.f));
            const float y0f = floorf(fy), x0f = floorf(fx);
            const int y0 = (int)y0f, x0 = (int)x0f;
            const float wy1 = fy - y0f, wy0 = (y0f + 1.f) - fy;
            const float wx1 = fx - x0f, wx0 = (x0f + 1.f) - fx;
            const int y1 = min(y0 + 1, p.H - 1), x1 = min(x0 + 1, p.W - 1);     // weights of clamped taps are 0
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* img = p.x_lr + (size_t)c * p.H * p.W;
                v[c] += img[(size_t)y0 * p.W + x0] * (wx0 * wy0) + img[(size_t)y0 * p.W + x1] * (wx1 * wy0) +
                        img[(size_t)y1 * p.W + x0] * (wx0 * wy1) + img[(size_t)y1 * p.W + x1] * (wx1 * wy1);
            }
        }
        p.rgb[q * 3] = v[0];
        p.rgb[q * 3 + 1] = v[1];
        p.rgb[q * 3 + 2] = v[2];
    }
}

// TAIL = k-steps of the ragged last chunk rounded up to {0: none, 2, 4, 8, 16}
template <typename M, int TAIL>
__global__ __launch_bounds__(XNT) void head_decode_fused_wide_kernel(FusedQP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* Xh = reinterpret_cast<unsigned short*>(smem_raw);
    unsigned short* Xl = reinterpret_cast<unsigned short*>(smem_raw + (M::AP ? M::ACT : 0));
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qbase = blockIdx.x * M::ROWS;

    f32x16 acc[M::MI];
    init_bias<M::MI>(acc, p.bias_in + 32 * w, lh);
    const __amdgpu_buffer_rsrc_t rs_z =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Z), 0, (unsigned)((size_t)p.nq * p.ldz * (M::AP ? 4 : 2)), 0x00020000);
    const unsigned w_bytes = 8u * (unsigned)p.nj_in * 1024u;
    const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.frag_in), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M::WP ? p.frag_in_lo : p.frag_in), 0, w_bytes, 0x00020000);
    const int k_full = p.Dv & ~(XH - 1);
#pragma unroll 1
    for (int k0 = 0; k0 < k_full; k0 += XH) decode_chunk<M, XKS>(Xh, Xl, p, rs_z, rs_wh, rs_wl, k0, XH, qbase, t, w, lane, acc);
    if (TAIL > 0) decode_chunk<M, (TAIL > 0 ? TAIL : 2)>(Xh, Xl, p, rs_z, rs_wh, rs_wl, k_full, p.Dv - k_full, qbase, t, w, lane, acc);
    __syncthreads();
    store_relu<M>(Xh, Xl, acc, 32 * w, li, lh);
    for (int l = 0; l < p.n_hidden; ++l) hidden_layer<M>(Xh, Xl, p.frag_hidden[l], p.frag_hidden_lo[l], p.bias_hidden[l], w, lane);
    __syncthreads();
    decode_tail<M>(Xh, Xl, p, t, qbase);
}

// ---- host side ----------------------------------------------------------------------------------
template <typename M>
static int launch_kv(const FusedKVP& p, hipStream_t s, const char* tag) {
    CIAOSR_BIG_LDS(head_kv_fused_wide_kernel<M>, M::KV_LDS);
    ProfScope prof(tag, s);
    hipLaunchKernelGGL(head_kv_fused_wide_kernel<M>, dim3(ceil_div(p.nq, M::QW)), dim3(XNT), M::KV_LDS, s, p);
    return launch_status(tag);
}

template <typename M>
static int launch_decode(const FusedQP& p, hipStream_t s, const char* tag) {
    const size_t lds = M::ACTS;
    CIAOSR_BIG_LDS((head_decode_fused_wide_kernel<M, 0>), lds);
    CIAOSR_BIG_LDS((head_decode_fused_wide_kernel<M, 2>), lds);
    CIAOSR_BIG_LDS((head_decode_fused_wide_kernel<M, 4>), lds);
    CIAOSR_BIG_LDS((head_decode_fused_wide_kernel<M, 8>), lds);
    CIAOSR_BIG_LDS((head_decode_fused_wide_kernel<M, 16>), lds);
    ProfScope prof(tag, s);
    const dim3 grid(ceil_div(p.nq, M::ROWS));
    const int tail_steps = ((p.Dv & (XH - 1)) + 15) >> 4;      // k-steps of the ragged last chunk (C = 64: 8, C = 180: 1)
    if (tail_steps == 0)
        hipLaunchKernelGGL((head_decode_fused_wide_kernel<M, 0>), grid, dim3(XNT), lds, s, p);
    else if (tail_steps <= 2)
        hipLaunchKernelGGL((head_decode_fused_wide_kernel<M, 2>), grid, dim3(XNT), lds, s, p);
    else if (tail_steps <= 4)
        hipLaunchKernelGGL((head_decode_fused_wide_kernel<M, 4>), grid, dim3(XNT), lds, s, p);
    else if (tail_steps <= 8)
        hipLaunchKernelGGL((head_decode_fused_wide_kernel<M, 8>), grid, dim3(XNT), lds, s, p);
    else
        hipLaunchKernelGGL((head_decode_fused_wide_kernel<M, 16>), grid, dim3(XNT), lds, s, p);
    return launch_status(tag);
}

// mode: 0 = f16 (half weights, half activations), 1 = f16-pairs (weight pairs), 2 = f16x3 (weight and activation pairs, fp32 Z)
int head_kv_fused_wide(const FusedKVP& p, int mode, hipStream_t s) {
    if (mode == 2) return launch_kv<ModeX3>(p, s, "head_kv_fused" CIAOSR_H16_SUFFIX "x3");
    if constexpr (kF16) {
        if (mode == 1) return launch_kv<ModePairs>(p, s, "head_kv_fused_f16");
        return launch_kv<ModeF16>(p, s, "head_kv_fused_f16");
    }
    return CIAOSR_ERR_UNSUPPORTED;       // the bf16 build carries the pair mode only
}

int head_decode_fused_wide(const FusedQP& p, int mode, hipStream_t s) {
    if (mode == 2) return launch_decode<ModeX3>(p, s, "head_decode_fused" CIAOSR_H16_SUFFIX "x3");
    if constexpr (kF16) {
        if (mode == 1) return launch_decode<ModePairs>(p, s, "head_decode_fused_f16");
        return launch_decode<ModeF16>(p, s, "head_decode_fused_f16");
    }
    return CIAOSR_ERR_UNSUPPORTED;
}

}  // namespace wide
}  // namespace CIAOSR_H16_NS
}  // namespace ciaosr

#if defined(CIAOSR_PROBE) && CIAOSR_F16
extern "C" int ciaosr_debug_probe_x3_read(unsigned long long* host, int n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ciaosr::f16::wide::g_xprobe), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
#endif
