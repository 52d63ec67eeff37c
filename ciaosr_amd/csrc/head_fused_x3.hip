// "f16x3": the fused head kernels with BOTH MFMA operands as IEEE-half hi + lo pairs -- the fp32-tolerance fast mode
// (ciaosr_options_t.f16_pairs = 2, hip_ops.Options('f16x3')).
//
// The f16-pairs mode (weights as pairs, half activations) lands 4 % outside the north star's |delta| <= 1e-3 on the full C3 tile, and
// what is left is the 11-bit rounding of the ACTIVATIONS of the three MLP chains (tools/pairs_probe.py).  Here an activation a is kept
// as hi = half(a), lo = half(a - hi) (~22 mantissa bits) and a product enters the fp32 accumulator as
//     w_hi a_hi + w_lo a_hi + w_hi a_lo          (three v_mfma_f32_32x32x16_f16; the dropped w_lo a_lo is ~2^-22 relative),
// i.e. 3/16 of the fp32 MFMA time per product.  Half subnormals (a lo half is subnormal for |a| < 0.125) pass the matrix pipe
// unflushed (tools/ubench/f16_denorm.hip).  Z travels as fp32 between the two kernels.
//
// Re-cut, not re-compiled: two activation arrays [128][264] half are 132 KB of LDS = ONE workgroup per CU, so the workgroup has
// 8 waves (two per SIMD, each owning ONE 32-column tile of a 256-wide layer: 64 accumulator registers) and the weight stream per
// MFMA drops -- a wave loads w_hi and w_lo (2 KB) for 12 MFMAs where the pairs kernel loads 2 KB for 8 and the plain kernel 1 KB
// for 4: the L2 -> CU operand budget that caps the 16-bit head allows ~0.68 MFMA-busy here against 0.45.
// Same row order (row m = 32 j + q), same swapped operands and epilogue math as head_fused_h16.hip.
#define CIAOSR_F16 1
#include "h16_util.h"
#include "index_math.h"
#include "ops.h"

namespace ciaosr {
namespace x3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifdef CIAOSR_PROBE      // developer probe build (make probe; tools/head_probe.py 192 f16x3): cycle stamps of workgroup phases
__device__ unsigned long long g_xprobe[4096 * 16];
#define XPROBE(slot) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_xprobe[blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define XPROBE(slot) do { } while (0)
#endif

constexpr int XBM = 128;           // rows per workgroup = 32 queries x 4 key samples (decode: 128 queries)
constexpr int XMI = XBM / 32;      // 32-row MFMA tiles
constexpr int XH = 256;            // hidden width
constexpr int XLD = XH + 8;        // LDS row stride in halves (528 B: conflict-free ds_read_b128)
constexpr int XKS = XH / 16;       // k-steps per 256-wide layer
constexpr int XNW = 8;             // waves per workgroup
constexpr int XNT = XNW * 64;      // threads
constexpr unsigned kOobX = 0xFFFFFFF0u;

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
    const float floor_ = RELU ? 0.f : -kHalfMax;
    const f32x2_t v = {__builtin_amdgcn_fmed3f(a, floor_, kHalfMax), __builtin_amdgcn_fmed3f(b, floor_, kHalfMax)};
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);
    // residual x - hi (exact in fp32) as ONE v_fma_mix_f32 per element (fma of the half operand, taken straight from its half of the
    // packed register, with -1.0 and the fp32 x) instead of a convert and a subtract: 6 VALU per pair instead of 8 in an epilogue that
    // no MFMA covers (hipcc folds the fma(x, -1, y) form back into cvt + sub, hence the asm)
    const unsigned hp = __builtin_bit_cast(unsigned, h);
    f32x2_t r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(hp), "v"(v.x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(hp), "v"(v.y));
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}
// hi + lo of packed element 0 / 1 back to fp32
// (one v_fma_mix_f32 each: hi * 1.0 + lo with both half operands read from their packed registers)
__device__ __forceinline__ float pair0(unsigned hi, unsigned lo) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi), "v"(lo));
    return r;
}
__device__ __forceinline__ float pair1(unsigned hi, unsigned lo) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi), "v"(lo));
    return r;
}

__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) { return mfma_h16<true>(a, b, c); }

// first two weight stages of a pass: requested early so that they are in flight across the barrier in front of the pass
__device__ __forceinline__ void load_w01(const uint4* __restrict__ wh, const uint4* __restrict__ wl, uint4 (&fh)[3], uint4 (&fl)[3]) {
    fh[0] = wh[0]; fl[0] = wl[0];
    fh[1] = wh[64]; fl[1] = wl[64];
}

// acc[mi] += (W_hi + W_lo) . (X_hi + X_lo)^T - W_lo . X_lo^T over NKS (>= 2) k-steps of one 32-column weight tile.
// xh / xl: &X[lane row][8 g]; wh / wl: fragment streams (+ lane); fh / fl hold stages 0 and 1 already.
template <int NKS>
__device__ __forceinline__ void mma_pass_x3(const unsigned short* xh, const unsigned short* xl, const uint4* __restrict__ wh,
                                            const uint4* __restrict__ wl, f32x16 (&acc)[XMI], uint4 (&fh)[3], uint4 (&fl)[3]) {
    uint4 ah[2][XMI], al[2][XMI];
#pragma unroll
    for (int mi = 0; mi < XMI; ++mi) {
        ah[0][mi] = *reinterpret_cast<const uint4*>(xh + mi * 32 * XLD);
        al[0][mi] = *reinterpret_cast<const uint4*>(xl + mi * 32 * XLD);
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 2 < NKS) {
            fh[(ks + 2) % 3] = wh[(long)(ks + 2) * 64];
            fl[(ks + 2) % 3] = wl[(long)(ks + 2) * 64];
        }
        if (ks + 1 < NKS) {
#pragma unroll
            for (int mi = 0; mi < XMI; ++mi) {
                ah[(ks + 1) & 1][mi] = *reinterpret_cast<const uint4*>(xh + mi * 32 * XLD + 16 * (ks + 1));
                al[(ks + 1) & 1][mi] = *reinterpret_cast<const uint4*>(xl + mi * 32 * XLD + 16 * (ks + 1));
            }
        }
        const uint4 w_hi = fh[ks % 3], w_lo = fl[ks % 3];
        // the small terms first, each product type over the four independent accumulators
#pragma unroll
        for (int mi = 0; mi < XMI; ++mi) acc[mi] = mfma(w_lo, ah[ks & 1][mi], acc[mi]);
#pragma unroll
        for (int mi = 0; mi < XMI; ++mi) acc[mi] = mfma(w_hi, al[ks & 1][mi], acc[mi]);
#pragma unroll
        for (int mi = 0; mi < XMI; ++mi) acc[mi] = mfma(w_hi, ah[ks & 1][mi], acc[mi]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// accumulators of the wave's 32-column tile initialised with the bias (channel 8 g + 4 lh + e of the tile <-> acc[mi][4 g + e])
__device__ __forceinline__ void init_bias(f32x16 (&acc)[XMI], const float* __restrict__ bias32, int lh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(bias32 + 8 * g + 4 * lh);
#pragma unroll
        for (int mi = 0; mi < XMI; ++mi) {
            acc[mi][4 * g] = b.x; acc[mi][4 * g + 1] = b.y; acc[mi][4 * g + 2] = b.z; acc[mi][4 * g + 3] = b.w;
        }
    }
}

// relu + split of the wave's accumulators into the two activation arrays, columns [col0, col0 + 32)
__device__ __forceinline__ void store_relu_split(unsigned short* Xh, unsigned short* Xl, const f32x16 (&acc)[XMI], int col0, int li, int lh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = col0 + 8 * g + 4 * lh;
#pragma unroll
        for (int mi = 0; mi < XMI; ++mi) {
            uint2 h, l;
            split2<true>(acc[mi][4 * g], acc[mi][4 * g + 1], h.x, l.x);
            split2<true>(acc[mi][4 * g + 2], acc[mi][4 * g + 3], h.y, l.y);
            *reinterpret_cast<uint2*>(Xh + (32 * mi + li) * XLD + col) = h;
            *reinterpret_cast<uint2*>(Xl + (32 * mi + li) * XLD + col) = l;
        }
    }
}

// hidden layer, in place.  Entry: X may still be being written by other waves (the leading barrier orders it); exit: X holds this
// layer's activations as far as THIS wave's stores go -- the next consumer starts with a barrier.
__device__ __forceinline__ void hidden_layer_x3(unsigned short* Xh, unsigned short* Xl, const void* __restrict__ frag,
                                                const void* __restrict__ frag_lo, const float* __restrict__ bias, int w, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const uint4* wh = reinterpret_cast<const uint4*>(frag) + (size_t)w * XKS * 64 + lane;
    const uint4* wl = reinterpret_cast<const uint4*>(frag_lo) + (size_t)w * XKS * 64 + lane;
    uint4 fh[3], fl[3];
    load_w01(wh, wl, fh, fl);
    f32x16 acc[XMI];
    init_bias(acc, bias + 32 * w, lh);
    __syncthreads();
    mma_pass_x3<XKS>(Xh + li * XLD + 8 * lh, Xl + li * XLD + 8 * lh, wh, wl, acc, fh, fl);
    __syncthreads();
    store_relu_split(Xh, Xl, acc, 32 * w, li, lh);
}

// layer-0 rows from the hoisted fp32 tables: 16 rows per thread, all 16 table gathers in flight at once
__device__ __forceinline__ void build_rows_x3(unsigned short* Xh, unsigned short* Xl, const FusedChain& c, const int* s_kpix, const float* s_t4,
                                              int t) {
    const int n4 = t & 63, r0 = t >> 6;
    float4 tw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) tw[e] = *reinterpret_cast<const float4*>(c.tail + (size_t)(4 * n4 + e) * c.ld_tail);
    constexpr int NR = XBM / XNW;
    float4 tv[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) tv[i] = reinterpret_cast<const float4*>(c.table + (size_t)s_kpix[r0 + XNW * i] * XH)[n4];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = r0 + XNW * i;
        const float4 q = *reinterpret_cast<const float4*>(s_t4 + 4 * r);       // rel_y rel_x scale_y scale_x
        uint2 h, l;
        split2<true>(tv[i].x + tw[0].x * q.x + tw[0].y * q.y + tw[0].z * q.z + tw[0].w * q.w,
                     tv[i].y + tw[1].x * q.x + tw[1].y * q.y + tw[1].z * q.z + tw[1].w * q.w, h.x, l.x);
        split2<true>(tv[i].z + tw[2].x * q.x + tw[2].y * q.y + tw[2].z * q.z + tw[2].w * q.w,
                     tv[i].w + tw[3].x * q.x + tw[3].y * q.y + tw[3].z * q.z + tw[3].w * q.w, h.y, l.y);
        *reinterpret_cast<uint2*>(Xh + r * XLD + 4 * n4) = h;
        *reinterpret_cast<uint2*>(Xl + r * XLD + 4 * n4) = l;
    }
}

constexpr size_t kXAct = (size_t)XBM * XLD * 2;      // one activation array
constexpr size_t kKvLds = 2 * kXAct + (size_t)(XBM * 4 + XNW * XBM + XBM) * sizeof(float) + (XBM + XBM / 4 + XBM) * sizeof(int);

__global__ __launch_bounds__(XNT) void head_kv_fused_x3_kernel(FusedKVP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* Xh = reinterpret_cast<unsigned short*>(smem_raw);                 // [128][264] half: hi
    unsigned short* Xl = reinterpret_cast<unsigned short*>(smem_raw + kXAct);         // [128][264] half: lo
    float* s_t4 = reinterpret_cast<float*>(smem_raw + 2 * kXAct);                     // [128][4]
    float* s_part = s_t4 + XBM * 4;                                                   // [8][128]
    float* s_attn = s_part + XNW * XBM;                                               // [128]
    int* s_kpix = reinterpret_cast<int*>(s_attn + XBM);                               // [128]
    int* s_qpix = s_kpix + XBM;                                                       // [32]
    int* s_goff = s_qpix + XBM / 4;                                                   // [128]

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qbase = blockIdx.x * (XBM / 4);

    XPROBE(0);
    // ---- index math: row m = 32 j + q (exactly head_fused_h16.hip's) ----------------------------------------------
    int bad = 0;
    if (t < XBM) {
        const int ql = qbase + (t & 31), j = t >> 5;
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
            if (j == 0) s_qpix[t] = qin ? iy * p.W + ix : -1;
            if (qin) {
                const int oy = s.ky - iy, ox = s.kx - ix;
                if (oy >= -1 && oy <= 1 && ox >= -1 && ox <= 1) goff = (iy * p.W + ix) * 9 + (oy + 1) * 3 + (ox + 1);
                else bad = 1;
            }
        } else if (j == 0) {
            s_qpix[t] = -1;
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
    build_rows_x3(Xh, Xl, p.k, s_kpix, s_t4, t);
    XPROBE(2);
    for (int l = 0; l < p.k.n_hidden; ++l) {
        hidden_layer_x3(Xh, Xl, p.k.frag_hidden[l], p.k.frag_hidden_lo[l], p.k.bias_hidden[l], w, lane);
        if (l < 3) XPROBE(10 + l);
    }
    __syncthreads();
    XPROBE(3);
    if (table) {
        // logit = h4 . G[query pixel, key offset] + c (fp32 table, hi + lo activations): 4 threads per row, 16 gathers in flight
        const int row = t >> 2, part = t & 3;
        const int go = s_goff[row];
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, p.g_bytes, 0x00020000);
        const unsigned gbase = go >= 0 ? (unsigned)go * (unsigned)p.ldg * 4u : kOobX;
        float a = (go >= 0 && part == 0) ? p.G[(size_t)go * p.ldg + 256] : 0.f;
        float4 gv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) gv[i] = xload4(rs_g, gbase == kOobX ? kOobX : gbase + (unsigned)(16 * i + 4 * part) * 4u);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint2 xh = *reinterpret_cast<const uint2*>(Xh + row * XLD + 16 * i + 4 * part);
            const uint2 xl = *reinterpret_cast<const uint2*>(Xl + row * XLD + 16 * i + 4 * part);
            a += pair0(xh.x, xl.x) * gv[i].x + pair1(xh.x, xl.x) * gv[i].y + pair0(xh.y, xl.y) * gv[i].z + pair1(xh.y, xl.y) * gv[i].w;
        }
        a += quad_xor1(a);
        a += quad_xor2(a);
        s_part[part * XBM + row] = part == 0 ? a : 0.f;
        s_part[(part + 4) * XBM + row] = 0.f;
    } else {
        // fallback (no table, or a key outside the query's 3x3 neighbourhood): imnet_k's output layer on the MFMA
        float part[XMI];
        unsigned koff[XMI];
#pragma unroll
        for (int mi = 0; mi < XMI; ++mi) {
            part[mi] = 0.f;
            koff[mi] = (unsigned)s_kpix[32 * mi + li] * (unsigned)p.ldu * 4u;
        }
        const int qp = s_qpix[li];
        const unsigned qoff = qp >= 0 ? (unsigned)qp * (unsigned)p.ldu * 4u : kOobX;
        const int n_units = (p.k.n_out + 31) >> 5;
        const __amdgpu_buffer_rsrc_t rs_bk =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.k.bias_out), 0, (unsigned)p.k.n_out * 4u, 0x00020000);
        for (int u = w; u < n_units; u += XNW) {
            const uint4* wh = reinterpret_cast<const uint4*>(p.k.frag_out) + (size_t)u * XKS * 64 + lane;
            const uint4* wl = reinterpret_cast<const uint4*>(p.k.frag_out_lo) + (size_t)u * XKS * 64 + lane;
            uint4 fh[3], fl[3];
            load_w01(wh, wl, fh, fl);
            f32x16 acc[XMI];
#pragma unroll
            for (int mi = 0; mi < XMI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
            mma_pass_x3<XKS>(Xh + li * XLD + 8 * lh, Xl + li * XLD + 8 * lh, wh, wl, acc, fh, fl);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * u + 8 * g + 4 * lh;
                const unsigned doff = d0 < p.k.n_out ? (unsigned)d0 * 4u : kOobX;
                const float4 bv = xload4(rs_bk, doff);
                const float4 qv = xload4(rs_u, (doff == kOobX || qoff == kOobX) ? kOobX : qoff + doff);
                float4 kv[XMI];
#pragma unroll
                for (int mi = 0; mi < XMI; ++mi) kv[mi] = xload4(rs_u, doff == kOobX ? kOobX : koff[mi] + doff);
#pragma unroll
                for (int mi = 0; mi < XMI; ++mi)
                    part[mi] += qv.x * (kv[mi].x * (acc[mi][4 * g] + bv.x)) + qv.y * (kv[mi].y * (acc[mi][4 * g + 1] + bv.y)) +
                                qv.z * (kv[mi].z * (acc[mi][4 * g + 2] + bv.z)) + qv.w * (kv[mi].w * (acc[mi][4 * g + 3] + bv.w));
            }
        }
#pragma unroll
        for (int mi = 0; mi < XMI; ++mi) {
            part[mi] += __shfl_xor(part[mi], 32, 64);
            if (lh == 0) s_part[w * XBM + 32 * mi + li] = part[mi];
        }
    }
    __syncthreads();
    if (t < XBM / 4) {          // 4-way softmax of query t: rows t, t + 32, t + 64, t + 96
        float lg[4], m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 32 * j + t;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < XNW; ++k) sum += s_part[k * XBM + row];
            lg[j] = sum / p.softmax_scale;
            m = fmaxf(m, lg[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { lg[j] = expf(lg[j] - m); den += lg[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) s_attn[32 * j + t] = lg[j] / den;
    }

    // ================= phi_v =====================================================================
    XPROBE(4);
    build_rows_x3(Xh, Xl, p.v, s_kpix, s_t4, t);          // every wave is past its logit reads of X (barrier above)
    XPROBE(5);
    for (int l = 0; l < p.v.n_hidden; ++l) {
        hidden_layer_x3(Xh, Xl, p.v.frag_hidden[l], p.v.frag_hidden_lo[l], p.v.bias_hidden[l], w, lane);
        if (l < 3) XPROBE(13 + l);
    }
    __syncthreads();
    XPROBE(6);
    {
        const int n_units = (p.v.n_out + 31) >> 5;
        const __amdgpu_buffer_rsrc_t rs_bv =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.v.bias_out), 0, (unsigned)p.v.n_out * 4u, 0x00020000);
        // Z rows are fp32: [nq][ldz]
        const __amdgpu_buffer_rsrc_t rs_z =
            __builtin_amdgcn_make_buffer_rsrc(p.Z, 0, (unsigned)((size_t)p.nq * p.ldz * 4), 0x00020000);
        unsigned voff[XMI];
        float av[XMI];
#pragma unroll
        for (int mi = 0; mi < XMI; ++mi) {
            voff[mi] = (unsigned)s_kpix[32 * mi + li] * (unsigned)p.ldu * 4u;
            av[mi] = s_attn[32 * mi + li];
        }
        const int ql = qbase + li;
        const unsigned zoff = ql < p.nq ? (unsigned)ql * (unsigned)p.ldz * 4u : kOobX;
        for (int u = w; u < n_units; u += XNW) {
            const uint4* wh = reinterpret_cast<const uint4*>(p.v.frag_out) + (size_t)u * XKS * 64 + lane;
            const uint4* wl = reinterpret_cast<const uint4*>(p.v.frag_out_lo) + (size_t)u * XKS * 64 + lane;
            uint4 fh[3], fl[3];
            load_w01(wh, wl, fh, fl);
            // the unit's value rows and bias: in flight under the 192 MFMAs below
            float4 vv[4][XMI];
            f32x16 acc[XMI];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * u + 8 * g + 4 * lh;
                const unsigned doff = d0 < p.v.n_out ? (unsigned)d0 * 4u : kOobX;
                const float4 bv = xload4(rs_bv, doff);
#pragma unroll
                for (int mi = 0; mi < XMI; ++mi) {
                    vv[g][mi] = xload4(rs_u, doff == kOobX ? kOobX : voff[mi] + doff);
                    acc[mi][4 * g] = bv.x; acc[mi][4 * g + 1] = bv.y; acc[mi][4 * g + 2] = bv.z; acc[mi][4 * g + 3] = bv.w;
                }
            }
            mma_pass_x3<XKS>(Xh + li * XLD + 8 * lh, Xl + li * XLD + 8 * lh, wh, wl, acc, fh, fl);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int mi = 0; mi < XMI; ++mi) {
                    z.x = fmaf(av[mi] * vv[g][mi].x, acc[mi][4 * g], z.x);
                    z.y = fmaf(av[mi] * vv[g][mi].y, acc[mi][4 * g + 1], z.y);
                    z.z = fmaf(av[mi] * vv[g][mi].z, acc[mi][4 * g + 2], z.z);
                    z.w = fmaf(av[mi] * vv[g][mi].w, acc[mi][4 * g + 3], z.w);
                }
                const int d0 = 32 * u + 8 * g + 4 * lh;
                xstore4(rs_z, (zoff == kOobX || d0 >= p.v.n_out) ? kOobX : zoff + (unsigned)d0 * 4u, z);
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
// decode: Z arrives as fp32 rows; the input layer is streamed through LDS in 256-column chunks, split into hi + lo on the way in.
// One chunk: [weight prefetch] [barrier] [128 x 256 piece of Z -> hi / lo in LDS] [barrier] [NKS k-steps].  Columns >= kc are zero in
// LDS and the weight stream is read through a bounded buffer descriptor (0 beyond its end): a ragged last chunk may run whole
// k-steps past Dv.
template <int NKS>
__device__ __forceinline__ void decode_chunk_x3(unsigned short* Xh, unsigned short* Xl, const FusedQP& p, __amdgpu_buffer_rsrc_t rs_z,
                                                __amdgpu_buffer_rsrc_t rs_wh, __amdgpu_buffer_rsrc_t rs_wl, int k0, int kc, int qbase, int t,
                                                int w, int lane, f32x16 (&acc)[XMI]) {
    const int li = lane & 31, lh = lane >> 5;
    const unsigned wbase = ((unsigned)w * (unsigned)p.nj_in + (unsigned)(k0 >> 4)) * 64u * 16u + (unsigned)lane * 16u;
    auto wload = [&](__amdgpu_buffer_rsrc_t rs, int ks) -> uint4 {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(wbase + (unsigned)ks * 1024u), 0, 0);
        return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
    };
    uint4 fh[3], fl[3];
    fh[0] = wload(rs_wh, 0); fl[0] = wload(rs_wl, 0);
    fh[1] = wload(rs_wh, 1); fl[1] = wload(rs_wl, 1);
    if (k0 > 0) __syncthreads();
    {
        const int c4 = (t & 63) * 4, r0 = t >> 6;
        float4 zv[XBM / XNW];
#pragma unroll
        for (int i = 0; i < XBM / XNW; ++i) {
            const int ql = qbase + r0 + XNW * i;
            zv[i] = xload4(rs_z, (ql < p.nq && c4 < kc) ? ((unsigned)ql * (unsigned)p.ldz + (unsigned)(k0 + c4)) * 4u : kOobX);
        }
#pragma unroll
        for (int i = 0; i < XBM / XNW; ++i) {
            uint2 h, l;
            split2<false>(zv[i].x, zv[i].y, h.x, l.x);
            split2<false>(zv[i].z, zv[i].w, h.y, l.y);
            *reinterpret_cast<uint2*>(Xh + (r0 + XNW * i) * XLD + c4) = h;
            *reinterpret_cast<uint2*>(Xl + (r0 + XNW * i) * XLD + c4) = l;
        }
    }
    __syncthreads();
    const unsigned short* xh = Xh + li * XLD + 8 * lh;
    const unsigned short* xl = Xl + li * XLD + 8 * lh;
    uint4 ah[2][XMI], al[2][XMI];
#pragma unroll
    for (int mi = 0; mi < XMI; ++mi) {
        ah[0][mi] = *reinterpret_cast<const uint4*>(xh + mi * 32 * XLD);
        al[0][mi] = *reinterpret_cast<const uint4*>(xl + mi * 32 * XLD);
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 2 < NKS) {
            fh[(ks + 2) % 3] = wload(rs_wh, ks + 2);
            fl[(ks + 2) % 3] = wload(rs_wl, ks + 2);
        }
        if (ks + 1 < NKS) {
#pragma unroll
            for (int mi = 0; mi < XMI; ++mi) {
                ah[(ks + 1) & 1][mi] = *reinterpret_cast<const uint4*>(xh + mi * 32 * XLD + 16 * (ks + 1));
                al[(ks + 1) & 1][mi] = *reinterpret_cast<const uint4*>(xl + mi * 32 * XLD + 16 * (ks + 1));
            }
        }
        const uint4 w_hi = fh[ks % 3], w_lo = fl[ks % 3];
#pragma unroll
        for (int mi = 0; mi < XMI; ++mi) acc[mi] = mfma(w_lo, ah[ks & 1][mi], acc[mi]);
#pragma unroll
        for (int mi = 0; mi < XMI; ++mi) acc[mi] = mfma(w_hi, al[ks & 1][mi], acc[mi]);
#pragma unroll
        for (int mi = 0; mi < XMI; ++mi) acc[mi] = mfma(w_hi, ah[ks & 1][mi], acc[mi]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// last Linear (256 -> 3) in fp32 on the hi + lo activations + bilinear/border residual (net:107-108,221): 4 threads per row
__device__ __forceinline__ void decode_tail_x3(const unsigned short* Xh, const unsigned short* Xl, const FusedQP& p, int t, int qbase) {
    const int row = t >> 2, part = t & 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    {
        const unsigned short* xh = Xh + row * XLD + 64 * part;
        const unsigned short* xl = Xl + row * XLD + 64 * part;
        const float* w0 = p.w_last + 64 * part;
        const float* w1 = w0 + p.ld_last;
        const float* w2 = w1 + p.ld_last;
#pragma unroll 4
        for (int n = 0; n < 64; n += 4) {
            const uint2 hb = *reinterpret_cast<const uint2*>(xh + n);
            const uint2 lb = *reinterpret_cast<const uint2*>(xl + n);
            const float x0 = pair0(hb.x, lb.x), x1 = pair1(hb.x, lb.x), x2 = pair0(hb.y, lb.y), x3 = pair1(hb.y, lb.y);
            const float4 u0 = *reinterpret_cast<const float4*>(w0 + n);
            const float4 u1 = *reinterpret_cast<const float4*>(w1 + n);
            const float4 u2 = *reinterpret_cast<const float4*>(w2 + n);
            a0 += x0 * u0.x + x1 * u0.y + x2 * u0.z + x3 * u0.w;
            a1 += x0 * u1.x + x1 * u1.y + x2 * u1.z + x3 * u1.w;
            a2 += x0 * u2.x + x1 * u2.y + x2 * u2.z + x3 * u2.w;
        }
    }
    a0 += quad_xor1(a0); a0 += quad_xor2(a0);
    a1 += quad_xor1(a1); a1 += quad_xor2(a1);
    a2 += quad_xor1(a2); a2 += quad_xor2(a2);
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
template <int TAIL>
__global__ __launch_bounds__(XNT) void head_decode_fused_x3_kernel(FusedQP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* Xh = reinterpret_cast<unsigned short*>(smem_raw);
    unsigned short* Xl = reinterpret_cast<unsigned short*>(smem_raw + kXAct);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qbase = blockIdx.x * XBM;

    f32x16 acc[XMI];
    init_bias(acc, p.bias_in + 32 * w, lh);
    const __amdgpu_buffer_rsrc_t rs_z =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Z), 0, (unsigned)((size_t)p.nq * p.ldz * 4), 0x00020000);
    const unsigned w_bytes = 8u * (unsigned)p.nj_in * 1024u;
    const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.frag_in), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.frag_in_lo), 0, w_bytes, 0x00020000);
    const int k_full = p.Dv & ~(XH - 1);
#pragma unroll 1
    for (int k0 = 0; k0 < k_full; k0 += XH) decode_chunk_x3<XKS>(Xh, Xl, p, rs_z, rs_wh, rs_wl, k0, XH, qbase, t, w, lane, acc);
    if (TAIL > 0) decode_chunk_x3<(TAIL > 0 ? TAIL : 2)>(Xh, Xl, p, rs_z, rs_wh, rs_wl, k_full, p.Dv - k_full, qbase, t, w, lane, acc);
    __syncthreads();
    store_relu_split(Xh, Xl, acc, 32 * w, li, lh);
    for (int l = 0; l < p.n_hidden; ++l) hidden_layer_x3(Xh, Xl, p.frag_hidden[l], p.frag_hidden_lo[l], p.bias_hidden[l], w, lane);
    __syncthreads();
    decode_tail_x3(Xh, Xl, p, t, qbase);
}

// ---- host side ----------------------------------------------------------------------------------
int head_kv_fused_x3(const FusedKVP& p, hipStream_t s) {
    CIAOSR_BIG_LDS(head_kv_fused_x3_kernel, kKvLds);
    ProfScope prof("head_kv_fused_f16x3", s);
    hipLaunchKernelGGL(head_kv_fused_x3_kernel, dim3(ceil_div(p.nq, XBM / 4)), dim3(XNT), kKvLds, s, p);
    return launch_status("head_kv_fused_f16x3");
}

int head_decode_fused_x3(const FusedQP& p, hipStream_t s) {
    const size_t lds = 2 * kXAct;
    CIAOSR_BIG_LDS(head_decode_fused_x3_kernel<0>, lds);
    CIAOSR_BIG_LDS(head_decode_fused_x3_kernel<2>, lds);
    CIAOSR_BIG_LDS(head_decode_fused_x3_kernel<4>, lds);
    CIAOSR_BIG_LDS(head_decode_fused_x3_kernel<8>, lds);
    CIAOSR_BIG_LDS(head_decode_fused_x3_kernel<16>, lds);
    ProfScope prof("head_decode_fused_f16x3", s);
    const dim3 grid(ceil_div(p.nq, XBM));
    const int tail_steps = ((p.Dv & (XH - 1)) + 15) >> 4;      // k-steps of the ragged last chunk (C = 64: 8, C = 180: 1)
    if (tail_steps == 0)
        hipLaunchKernelGGL(head_decode_fused_x3_kernel<0>, grid, dim3(XNT), lds, s, p);
    else if (tail_steps <= 2)
        hipLaunchKernelGGL(head_decode_fused_x3_kernel<2>, grid, dim3(XNT), lds, s, p);
    else if (tail_steps <= 4)
        hipLaunchKernelGGL(head_decode_fused_x3_kernel<4>, grid, dim3(XNT), lds, s, p);
    else if (tail_steps <= 8)
        hipLaunchKernelGGL(head_decode_fused_x3_kernel<8>, grid, dim3(XNT), lds, s, p);
    else
        hipLaunchKernelGGL(head_decode_fused_x3_kernel<16>, grid, dim3(XNT), lds, s, p);
    return launch_status("head_decode_fused_f16x3");
}

}  // namespace x3
}  // namespace ciaosr

#ifdef CIAOSR_PROBE
extern "C" int ciaosr_debug_probe_x3_read(unsigned long long* host, int n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ciaosr::x3::g_xprobe), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
#endif
