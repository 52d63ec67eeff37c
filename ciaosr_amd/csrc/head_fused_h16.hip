// 16-bit-MFMA variant of the fused head kernels (precision modes "bf16" and "f16"; this file is compiled once per element
// type, h16_util.h): MFMA inputs 16-bit, fp32 accumulate; coordinates, index math, layer-0 tables, logits, softmax, the
// attention-weighted sum and the decode output stay fp32.  v_mfma_f32_32x32x16_{bf16,f16} runs 16x
// the fp32 MFMA rate, which moves the kernel from the MFMA roofline to the per-CU weight stream out of L2, so the row
// tile doubles to 128 rows (= 32 queries x 4 key samples) per workgroup: 16-bit activations [128][264] are 66 KB, two
// workgroups still fit a CU, and every weight fragment fetched is used for twice as many rows.
// ("bf16" in the comments below = the 16-bit element type of the build.)
//
// Same structure as head_fused.hip: swapped MFMA operands (weights = A operand from L2 in pre-packed fragment
// order [n_tile][k16][lane][8 bf16], activations = B operand from LDS via ds_read_b128), a lane owns one
// activation row and 4x4 consecutive output channels, hidden layers in place in LDS.
#include "h16_util.h"
#include "index_math.h"
#include "ops.h"

namespace ciaosr {
namespace CIAOSR_H16_NS {

constexpr bool kF16 = CIAOSR_F16 != 0;

#ifdef CIAOSR_PROBE      // developer probe build (make probe; tools/head_probe.py bf16): cycle stamps of workgroup phases
__device__ unsigned long long g_hprobe16[4096 * 16];
#define HPROBE16(slot) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_hprobe16[blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define HPROBE16(slot) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int HBM_ = 128;          // rows per workgroup
constexpr int HMI = HBM_ / 32;     // 32-row MFMA tiles per workgroup
constexpr int HH = 256;            // hidden width
constexpr int HLD = HH + 8;        // LDS row stride in bf16 (528 B: conflict-free ds_read_b128)
constexpr int HKS = HH / 16;       // k-steps of 16 per 256-wide layer
constexpr unsigned kOobH = 0xFFFFFFF0u;

__device__ __forceinline__ unsigned pack2(float a, float b) { return pack_h16x2<kF16>(a, b); }
__device__ __forceinline__ unsigned pack_relu2(float a, float b) { return pack_relu_h16x2<kF16>(a, b); }

__device__ __forceinline__ float4 hload4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ void hstore4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, float4 v) {
    i32x4 iv;
    iv.x = __float_as_int(v.x); iv.y = __float_as_int(v.y); iv.z = __float_as_int(v.z); iv.w = __float_as_int(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(iv, rsrc, (int)byte_off, 0, 0);
}

// ---- fragment packing: W [N][ld] fp32 (K valid columns) -> P[nt][ks][lane][8 bf16]; lane (i = lane&31,
// g = lane>>5) holds W[32nt + i][16ks + 8g .. 16ks + 8g + 7]; zero padded.
// residual != 0: the element packed is w - bf16(w) (the low half of the hi + lo pair) instead of w
// P_lo != null: the residual form goes there as well (one launch packs the hi + lo pair)
__global__ void pack_fragments_h16_kernel(const float* __restrict__ W, int ld, int N, int K, uint4* __restrict__ P,
                                           int n_tiles, int nks, int residual, uint4* __restrict__ P_lo = nullptr) {
    const long total = (long)n_tiles * nks * 64;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long t = idx >> 6;
        const int ks = (int)(t % nks), nt = (int)(t / nks);
        const int n = nt * 32 + (lane & 31), k = 16 * ks + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = (n < N && k + e < K) ? W[(size_t)n * ld + k + e] : 0.f;
            if (residual) v[e] -= h16_lo<kF16>(to_h16<kF16>(v[e]));
        }
        P[idx] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
        if (P_lo) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] -= h16_lo<kF16>(to_h16<kF16>(v[e]));
            P_lo[idx] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
        }
    }
}

// acc[mi][ni] (+)= W_tile . X_tile^T over nks k-steps.  xa: &X[lane row][8g] (bf16), wf: fragment stream (+lane)
template <int NT>
__device__ __forceinline__ void mma_pass16(const unsigned short* xa, const uint4* __restrict__ wf, int nks, long tile_stride,
                                           f32x16 (&acc)[HMI][NT]) {
    // software pipeline: weight fragments (L2) are requested two k-steps ahead, the activation fragments (LDS) one
    // k-step ahead, so neither latency sits between two MFMA groups
    uint4 fb0[NT], fb1[NT], fb2[NT];
    uint4 fa0[HMI], fa1[HMI];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        fb0[ni] = wf[ni * tile_stride];
        fb1[ni] = wf[ni * tile_stride + (nks > 1 ? 64 : 0)];
    }
#pragma unroll
    for (int mi = 0; mi < HMI; ++mi) fa0[mi] = *reinterpret_cast<const uint4*>(xa + mi * 32 * HLD);
#pragma unroll 1
    for (int ks = 0; ks < nks; ++ks) {
        if (ks + 2 < nks) {
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) fb2[ni] = wf[ni * tile_stride + (long)(ks + 2) * 64];
        }
        if (ks + 1 < nks) {
#pragma unroll
            for (int mi = 0; mi < HMI; ++mi)
                fa1[mi] = *reinterpret_cast<const uint4*>(xa + mi * 32 * HLD + 16 * (ks + 1));
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const uint4 w = fb0[ni];
#pragma unroll
            for (int mi = 0; mi < HMI; ++mi) acc[mi][ni] = mfma_h16<kF16>(w, fa0[mi], acc[mi][ni]);
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) { fb0[ni] = fb1[ni]; fb1[ni] = fb2[ni]; }
#pragma unroll
        for (int mi = 0; mi < HMI; ++mi) fa0[mi] = fa1[mi];
    }
}

template <int NT>
__device__ __forceinline__ void zero_acc16(f32x16 (&acc)[HMI][NT]) {
#pragma unroll
    for (int mi = 0; mi < HMI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
}

// last Linear (256 -> 3) in fp32 on the bf16 activations + bilinear/border residual (net:107-108,221)
__device__ __forceinline__ void decode_tail16(const unsigned short* X, const FusedQP& p, int t, int qbase) {
    const int row = t >> 1, part = t & 1;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    {
        const unsigned short* xr = X + row * HLD + 128 * part;
        const float* w0 = p.w_last + 128 * part;
        const float* w1 = w0 + p.ld_last;
        const float* w2 = w1 + p.ld_last;
#pragma unroll 4
        for (int n = 0; n < 128; n += 4) {
            const uint2 xb = *reinterpret_cast<const uint2*>(xr + n);
            const float x0 = h16_lo<kF16>(xb.x), x1 = h16_hi<kF16>(xb.x);
            const float x2 = h16_lo<kF16>(xb.y), x3 = h16_hi<kF16>(xb.y);
            const float4 u0 = *reinterpret_cast<const float4*>(w0 + n);
            const float4 u1 = *reinterpret_cast<const float4*>(w1 + n);
            const float4 u2 = *reinterpret_cast<const float4*>(w2 + n);
            a0 += x0 * u0.x + x1 * u0.y + x2 * u0.z + x3 * u0.w;
            a1 += x0 * u1.x + x1 * u1.y + x2 * u1.z + x3 * u1.w;
            a2 += x0 * u2.x + x1 * u2.y + x2 * u2.z + x3 * u2.w;
        }
    }
    a0 += quad_xor1(a0);
    a1 += quad_xor1(a1);
    a2 += quad_xor1(a2);
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

// =================================================================================================================
// The kernels.  128 rows = 32 queries x 4 key samples per workgroup, two workgroups per CU, weights = A operand straight
// from L2, activations = B operand from LDS.  Round 2 re-cut them after the SQ counters of the round-1 version showed the
// VALU -- not the matrix pipe -- as the busiest issue port (11-14 VALU instructions per MFMA, MFMA pipe 28 % busy):
//   * rows are SAMPLE-major: row m = 32 j + q (q = query within the workgroup, j = key sample), so the four samples of a
//     query sit in the four m-tiles of ONE lane and the attention-weighted sum z = sum_j a_j value_j * w_v,j is four FMAs on
//     the lane's own accumulators -- no DPP quad reductions, no selects (384 -> 144 VALU per 32-column unit);
//   * biases are the accumulators' initial values (the zeroing moves were there anyway): epilogues lose an add per element;
//   * the k-loop is fully unrolled over compile-time buffer indices (3 weight stages, 2 activation stages): no register
//     rotation moves (24 v_mov per k-step in round 1); a sched_barrier per k-step keeps hipcc from hoisting every load to the top;
//   * gathers are issued BEFORE the MFMA block that hides them: the 16 value-row float4 of a v-out unit before its 64
//     MFMAs, the first two weight fragments of a layer before the barrier that precedes it, table / logit-table rows 16
//     (not 8) at a time while no accumulator is live;
//   * Z leaves the kernel as bf16 (the decode kernel rounds it to bf16 before its first MFMA anyway: bit-identical
//     result, half the round trip).
// =================================================================================================================
template <int NT>
__device__ __forceinline__ void load_fb01(const uint4* __restrict__ wf, long tile_stride, uint4 (&fb)[3][NT]) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        fb[0][ni] = wf[ni * tile_stride];
        fb[1][ni] = wf[ni * tile_stride + 64];
    }
}

// acc[mi][ni] += W_tile . X_tile^T over NKS (>= 2) k-steps; fb[0], fb[1] hold the first two weight stages already
template <int NT, int NKS>
__device__ __forceinline__ void mma_pass16u(const unsigned short* xa, const uint4* __restrict__ wf, long tile_stride,
                                            f32x16 (&acc)[HMI][NT], uint4 (&fb)[3][NT]) {
    uint4 fa[2][HMI];
#pragma unroll
    for (int mi = 0; mi < HMI; ++mi) fa[0][mi] = *reinterpret_cast<const uint4*>(xa + mi * 32 * HLD);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 2 < NKS) {
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) fb[(ks + 2) % 3][ni] = wf[ni * tile_stride + (long)(ks + 2) * 64];
        }
        if (ks + 1 < NKS) {
#pragma unroll
            for (int mi = 0; mi < HMI; ++mi)
                fa[(ks + 1) & 1][mi] = *reinterpret_cast<const uint4*>(xa + mi * 32 * HLD + 16 * (ks + 1));
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const uint4 wv = fb[ks % 3][ni];
#pragma unroll
            for (int mi = 0; mi < HMI; ++mi) acc[mi][ni] = mfma_h16<kF16>(wv, fa[ks & 1][mi], acc[mi][ni]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// hidden layer, in place in LDS.  Entry: X may still be being written by other waves (the leading barrier orders it);
// exit: X holds this layer's activations as far as THIS wave's stores go -- the next consumer starts with a barrier.
__device__ __forceinline__ void hidden_layer16v2(unsigned short* X, const void* __restrict__ frag, const void* __restrict__ frag_lo,
                                                 const float* __restrict__ bias, int w, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const uint4* wf = reinterpret_cast<const uint4*>(frag) + (size_t)(2 * w) * HKS * 64 + lane;
    uint4 fb[3][2];
    load_fb01<2>(wf, (long)HKS * 64, fb);                     // in flight across the barrier
    f32x16 acc[HMI][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = *reinterpret_cast<const float4*>(bias + 64 * w + 32 * ni + 8 * g + 4 * lh);
#pragma unroll
            for (int mi = 0; mi < HMI; ++mi) {
                acc[mi][ni][4 * g] = b.x; acc[mi][ni][4 * g + 1] = b.y; acc[mi][ni][4 * g + 2] = b.z; acc[mi][ni][4 * g + 3] = b.w;
            }
        }
    __syncthreads();
    mma_pass16u<2, HKS>(X + li * HLD + 8 * lh, wf, (long)HKS * 64, acc, fb);
    if (frag_lo) {      // low halves of the weight pairs against the same activations (uniform branch)
        const uint4* wl = reinterpret_cast<const uint4*>(frag_lo) + (size_t)(2 * w) * HKS * 64 + lane;
        load_fb01<2>(wl, (long)HKS * 64, fb);
        mma_pass16u<2, HKS>(X + li * HLD + 8 * lh, wl, (long)HKS * 64, acc, fb);
    }
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = 64 * w + 32 * ni + 8 * g + 4 * lh;
#pragma unroll
            for (int mi = 0; mi < HMI; ++mi) {
                uint2 o;
                o.x = pack_relu2(acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1]);
                o.y = pack_relu2(acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]);
                *reinterpret_cast<uint2*>(X + (32 * mi + li) * HLD + col) = o;
            }
        }
}

// layer-0 rows from the hoisted tables: 32 rows per thread, all 32 table gathers in flight at once (no accumulator is live here:
// the registers are free, and the phase is bound by the gathers' latency, not by their bytes)
__device__ __forceinline__ void build_rows16v2(unsigned short* X, const FusedChain& c, const int* s_kpix, const float* s_t4, int t) {
    const int n4 = t & 63;
    float4 tw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) tw[e] = *reinterpret_cast<const float4*>(c.tail + (size_t)(4 * n4 + e) * c.ld_tail);
    constexpr int NR = HBM_ / 4;       // rows per thread
    float4 tv[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) tv[i] = reinterpret_cast<const float4*>(c.table + (size_t)s_kpix[(t >> 6) + 4 * i] * HH)[n4];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = (t >> 6) + 4 * i;
        const float4 q = *reinterpret_cast<const float4*>(s_t4 + 4 * r);       // rel_y rel_x scale_y scale_x
        uint2 o;
        o.x = pack_relu2(tv[i].x + tw[0].x * q.x + tw[0].y * q.y + tw[0].z * q.z + tw[0].w * q.w,
                         tv[i].y + tw[1].x * q.x + tw[1].y * q.y + tw[1].z * q.z + tw[1].w * q.w);
        o.y = pack_relu2(tv[i].z + tw[2].x * q.x + tw[2].y * q.y + tw[2].z * q.z + tw[2].w * q.w,
                         tv[i].w + tw[3].x * q.x + tw[3].y * q.y + tw[3].z * q.z + tw[3].w * q.w);
        *reinterpret_cast<uint2*>(X + r * HLD + 4 * n4) = o;
    }
}

__global__ __launch_bounds__(256, 2) void head_kv_fused_h16_kernel(FusedKVP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* X = reinterpret_cast<unsigned short*>(smem_raw);                 // [128][264] bf16
    float* s_t4 = reinterpret_cast<float*>(smem_raw + (size_t)HBM_ * HLD * 2);        // [128][4]
    float* s_part = s_t4 + HBM_ * 4;                                                  // [4][128]
    float* s_attn = s_part + 4 * HBM_;                                                // [128]
    int* s_kpix = reinterpret_cast<int*>(s_attn + HBM_);                              // [128]
    int* s_qpix = s_kpix + HBM_;                                                      // [32]
    int* s_goff = s_qpix + HBM_ / 4;                                                  // [128]

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qbase = blockIdx.x * (HBM_ / 4);
    if (p.gate && __builtin_nontemporal_load(p.gate) == 0) return;       // fallback launch behind the chained kernel: nothing to redo

    HPROBE16(0);
    // ---- index math: row m = 32 j + q --------------------------------------------------------------------------
    int bad = 0;
    if (t < HBM_) {
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
    HPROBE16(1);
    build_rows16v2(X, p.k, s_kpix, s_t4, t);
    HPROBE16(2);
    for (int l = 0; l < p.k.n_hidden; ++l) hidden_layer16v2(X, p.k.frag_hidden[l], p.k.frag_hidden_lo[l], p.k.bias_hidden[l], w, lane);
    __syncthreads();
    HPROBE16(3);
    if (table) {
        // logit = h4 . G[query pixel, key offset] + c (fp32 table, bf16 activations): 2 threads per row, 16 gathers in flight
        const int row = t >> 1, part = t & 1;
        const int go = s_goff[row];
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, p.g_bytes, 0x00020000);
        const unsigned gbase = go >= 0 ? (unsigned)go * (unsigned)p.ldg * 4u : kOobH;
        float a = (go >= 0 && part == 0) ? p.G[(size_t)go * p.ldg + 256] : 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float4 gv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) gv[i] = hload4(rs_g, gbase == kOobH ? kOobH : gbase + (unsigned)(8 * (16 * b + i) + 4 * part) * 4u);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint2 xb = *reinterpret_cast<const uint2*>(X + row * HLD + 8 * (16 * b + i) + 4 * part);
                a += h16_lo<kF16>(xb.x) * gv[i].x + h16_hi<kF16>(xb.x) * gv[i].y + h16_lo<kF16>(xb.y) * gv[i].z + h16_hi<kF16>(xb.y) * gv[i].w;
            }
        }
        a += quad_xor1(a);
        if (part == 0) {
            s_part[row] = a;
            s_part[HBM_ + row] = 0.f;
            s_part[2 * HBM_ + row] = 0.f;
            s_part[3 * HBM_ + row] = 0.f;
        }
    } else {
        // fallback (no table, or a key outside the query's 3x3 neighbourhood): imnet_k's output layer on the MFMA
        float part[HMI];
        unsigned koff[HMI];
#pragma unroll
        for (int mi = 0; mi < HMI; ++mi) {
            part[mi] = 0.f;
            koff[mi] = (unsigned)s_kpix[32 * mi + li] * (unsigned)p.ldu * 4u;
        }
        const int qp = s_qpix[li];
        const unsigned qoff = qp >= 0 ? (unsigned)qp * (unsigned)p.ldu * 4u : kOobH;
        const int n_units = (p.k.n_out + 31) >> 5;
        const __amdgpu_buffer_rsrc_t rs_bk =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.k.bias_out), 0, (unsigned)p.k.n_out * 4u, 0x00020000);
        for (int u = w; u < n_units; u += 4) {
            f32x16 acc[HMI][1];
            zero_acc16<1>(acc);
            mma_pass16<1>(X + li * HLD + 8 * lh, reinterpret_cast<const uint4*>(p.k.frag_out) + (size_t)u * HKS * 64 + lane, HKS,
                          0, acc);
            if (p.k.frag_out_lo)
                mma_pass16<1>(X + li * HLD + 8 * lh, reinterpret_cast<const uint4*>(p.k.frag_out_lo) + (size_t)u * HKS * 64 + lane, HKS,
                              0, acc);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * u + 8 * g + 4 * lh;
                const unsigned doff = d0 < p.k.n_out ? (unsigned)d0 * 4u : kOobH;
                const float4 bv = hload4(rs_bk, doff);
                const float4 qv = hload4(rs_u, (doff == kOobH || qoff == kOobH) ? kOobH : qoff + doff);
                float4 kv[HMI];
#pragma unroll
                for (int mi = 0; mi < HMI; ++mi) kv[mi] = hload4(rs_u, doff == kOobH ? kOobH : koff[mi] + doff);
#pragma unroll
                for (int mi = 0; mi < HMI; ++mi)
                    part[mi] += qv.x * (kv[mi].x * (acc[mi][0][4 * g] + bv.x)) + qv.y * (kv[mi].y * (acc[mi][0][4 * g + 1] + bv.y)) +
                                qv.z * (kv[mi].z * (acc[mi][0][4 * g + 2] + bv.z)) + qv.w * (kv[mi].w * (acc[mi][0][4 * g + 3] + bv.w));
            }
        }
#pragma unroll
        for (int mi = 0; mi < HMI; ++mi) {
            part[mi] += __shfl_xor(part[mi], 32, 64);
            if (lh == 0) s_part[w * HBM_ + 32 * mi + li] = part[mi];
        }
    }
    __syncthreads();
    if (t < HBM_ / 4) {          // 4-way softmax of query t: rows t, t + 32, t + 64, t + 96
        float lg[4], m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 32 * j + t;
            lg[j] = (s_part[row] + s_part[HBM_ + row] + s_part[2 * HBM_ + row] + s_part[3 * HBM_ + row]) / p.softmax_scale;
            m = fmaxf(m, lg[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { lg[j] = expf(lg[j] - m); den += lg[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) s_attn[32 * j + t] = lg[j] / den;
    }

    // ================= phi_v =====================================================================
    HPROBE16(4);
    build_rows16v2(X, p.v, s_kpix, s_t4, t);          // every wave is past its logit reads of X (barrier above)
    HPROBE16(5);
    for (int l = 0; l < p.v.n_hidden; ++l) hidden_layer16v2(X, p.v.frag_hidden[l], p.v.frag_hidden_lo[l], p.v.bias_hidden[l], w, lane);
    __syncthreads();
    HPROBE16(6);
    {
        const int n_units = (p.v.n_out + 31) >> 5;
        const __amdgpu_buffer_rsrc_t rs_bv =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.v.bias_out), 0, (unsigned)p.v.n_out * 4u, 0x00020000);
        // Z rows are bf16: [nq][ldz] unsigned short
        const __amdgpu_buffer_rsrc_t rs_z =
            __builtin_amdgcn_make_buffer_rsrc(p.Z, 0, (unsigned)((size_t)p.nq * p.ldz * 2), 0x00020000);
        unsigned voff[HMI];
        float av[HMI];
#pragma unroll
        for (int mi = 0; mi < HMI; ++mi) {
            voff[mi] = (unsigned)s_kpix[32 * mi + li] * (unsigned)p.ldu * 4u;
            av[mi] = s_attn[32 * mi + li];
        }
        const int ql = qbase + li;
        const unsigned zoff = ql < p.nq ? (unsigned)ql * (unsigned)p.ldz * 2u : kOobH;
        for (int u = w; u < n_units; u += 4) {
            const uint4* wf = reinterpret_cast<const uint4*>(p.v.frag_out) + (size_t)u * HKS * 64 + lane;
            uint4 fb[3][1];
            load_fb01<1>(wf, 0, fb);
            // the unit's value rows and bias: in flight under the 64 MFMAs below
            float4 vv[4][HMI];
            f32x16 acc[HMI][1];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * u + 8 * g + 4 * lh;
                const unsigned doff = d0 < p.v.n_out ? (unsigned)d0 * 4u : kOobH;
                const float4 bv = hload4(rs_bv, doff);
#pragma unroll
                for (int mi = 0; mi < HMI; ++mi) {
                    vv[g][mi] = hload4(rs_u, doff == kOobH ? kOobH : voff[mi] + doff);
                    acc[mi][0][4 * g] = bv.x; acc[mi][0][4 * g + 1] = bv.y; acc[mi][0][4 * g + 2] = bv.z; acc[mi][0][4 * g + 3] = bv.w;
                }
            }
            mma_pass16u<1, HKS>(X + li * HLD + 8 * lh, wf, 0, acc, fb);
            if (p.v.frag_out_lo) {
                const uint4* wl = reinterpret_cast<const uint4*>(p.v.frag_out_lo) + (size_t)u * HKS * 64 + lane;
                load_fb01<1>(wl, 0, fb);
                mma_pass16u<1, HKS>(X + li * HLD + 8 * lh, wl, 0, acc, fb);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int mi = 0; mi < HMI; ++mi) {
                    z.x = fmaf(av[mi] * vv[g][mi].x, acc[mi][0][4 * g], z.x);
                    z.y = fmaf(av[mi] * vv[g][mi].y, acc[mi][0][4 * g + 1], z.y);
                    z.z = fmaf(av[mi] * vv[g][mi].z, acc[mi][0][4 * g + 2], z.z);
                    z.w = fmaf(av[mi] * vv[g][mi].w, acc[mi][0][4 * g + 3], z.w);
                }
                const int d0 = 32 * u + 8 * g + 4 * lh;
                const uint2 zb = pack_h16x4<kF16>(z.x, z.y, z.z, z.w);
                typedef int i32x2 __attribute__((ext_vector_type(2)));
                i32x2 zi; zi.x = (int)zb.x; zi.y = (int)zb.y;
                __builtin_amdgcn_raw_buffer_store_b64(zi, rs_z, (int)((zoff == kOobH || d0 >= p.v.n_out) ? kOobH : zoff + (unsigned)d0 * 2u), 0, 0);
            }
        }
    }
    HPROBE16(7);
}

// ---------------------------------------------------------------------------------------------
// decode: Z arrives as bf16 rows; input layer streamed through LDS in 256-column chunks.
// One chunk: [weight prefetch] [barrier] [128 x 256 bf16 piece of Z -> LDS] [barrier] [NKS k-steps].  Columns >= kc are
// zero-filled in LDS, and the weight stream is read through a bounded buffer descriptor (0 beyond its end), so a ragged
// last chunk may run whole k-steps past Dv.
template <int NKS>
__device__ __forceinline__ void decode_chunk16(unsigned short* X, const FusedQP& p, __amdgpu_buffer_rsrc_t rs_z, __amdgpu_buffer_rsrc_t rs_w,
                                               __amdgpu_buffer_rsrc_t rs_wlo, bool has_lo, int k0, int kc, int qbase, int t, int w, int lane,
                                               f32x16 (&acc)[HMI][2]) {
    const int li = lane & 31, lh = lane >> 5;
    const unsigned tile_bytes = (unsigned)p.nj_in * 64u * 16u;
    const unsigned wbase = ((unsigned)(2 * w) * (unsigned)p.nj_in + (unsigned)(k0 >> 4)) * 64u * 16u + (unsigned)lane * 16u;
    auto wload = [&](int ni, int ks) -> uint4 {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wbase + (unsigned)ni * tile_bytes + (unsigned)ks * 1024u), 0, 0);
        return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
    };
    uint4 fb[3][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) { fb[0][ni] = wload(ni, 0); fb[1][ni] = wload(ni, 1); }
    if (k0 > 0) __syncthreads();
    {
        const int c8 = (t & 31) * 8;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            i32x4 zv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ql = qbase + (t >> 5) + 8 * (8 * b + i);
                zv[i] = __builtin_amdgcn_raw_buffer_load_b128(
                    rs_z, (int)((ql < p.nq && c8 < kc) ? ((unsigned)ql * (unsigned)p.ldz + (unsigned)(k0 + c8)) * 2u : kOobH), 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<i32x4*>(X + ((t >> 5) + 8 * (8 * b + i)) * HLD + c8) = zv[i];
        }
    }
    __syncthreads();
    const unsigned short* xa = X + li * HLD + 8 * lh;
    uint4 fa[2][HMI];
#pragma unroll
    for (int mi = 0; mi < HMI; ++mi) fa[0][mi] = *reinterpret_cast<const uint4*>(xa + mi * 32 * HLD);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 2 < NKS) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) fb[(ks + 2) % 3][ni] = wload(ni, ks + 2);
        }
        if (ks + 1 < NKS) {
#pragma unroll
            for (int mi = 0; mi < HMI; ++mi)
                fa[(ks + 1) & 1][mi] = *reinterpret_cast<const uint4*>(xa + mi * 32 * HLD + 16 * (ks + 1));
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const uint4 wv = fb[ks % 3][ni];
#pragma unroll
            for (int mi = 0; mi < HMI; ++mi) acc[mi][ni] = mfma_h16<kF16>(wv, fa[ks & 1][mi], acc[mi][ni]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (has_lo) {       // low halves of the weight pairs, same LDS chunk
        auto wload_lo = [&](int ni, int ks) -> uint4 {
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_wlo, (int)(wbase + (unsigned)ni * tile_bytes + (unsigned)ks * 1024u), 0, 0);
            return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
        };
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) { fb[0][ni] = wload_lo(ni, 0); fb[1][ni] = wload_lo(ni, 1); }
#pragma unroll
        for (int mi = 0; mi < HMI; ++mi) fa[0][mi] = *reinterpret_cast<const uint4*>(xa + mi * 32 * HLD);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 2 < NKS) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) fb[(ks + 2) % 3][ni] = wload_lo(ni, ks + 2);
            }
            if (ks + 1 < NKS) {
#pragma unroll
                for (int mi = 0; mi < HMI; ++mi)
                    fa[(ks + 1) & 1][mi] = *reinterpret_cast<const uint4*>(xa + mi * 32 * HLD + 16 * (ks + 1));
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const uint4 wv = fb[ks % 3][ni];
#pragma unroll
                for (int mi = 0; mi < HMI; ++mi) acc[mi][ni] = mfma_h16<kF16>(wv, fa[ks & 1][mi], acc[mi][ni]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// TAIL = k-steps of the ragged last chunk rounded up to {0: none, 2, 4, 8, 16}
template <int TAIL>
__global__ __launch_bounds__(256, 2) void head_decode_fused_h16_kernel(FusedQP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* X = reinterpret_cast<unsigned short*>(smem_raw);   // [128][264] bf16
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qbase = blockIdx.x * HBM_;

    f32x16 acc[HMI][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias_in + 64 * w + 32 * ni + 8 * g + 4 * lh);
#pragma unroll
            for (int mi = 0; mi < HMI; ++mi) {
                acc[mi][ni][4 * g] = b.x; acc[mi][ni][4 * g + 1] = b.y; acc[mi][ni][4 * g + 2] = b.z; acc[mi][ni][4 * g + 3] = b.w;
            }
        }
    const __amdgpu_buffer_rsrc_t rs_z =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Z), 0, (unsigned)((size_t)p.nq * p.ldz * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.frag_in), 0, (unsigned)(8u * (unsigned)p.nj_in * 1024u), 0x00020000);
    const bool has_lo = p.frag_in_lo != nullptr;
    const __amdgpu_buffer_rsrc_t rs_wlo = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(has_lo ? p.frag_in_lo : p.frag_in), 0, (unsigned)(8u * (unsigned)p.nj_in * 1024u), 0x00020000);
    const int k_full = p.Dv & ~(HH - 1);
#pragma unroll 1
    for (int k0 = 0; k0 < k_full; k0 += HH) decode_chunk16<HKS>(X, p, rs_z, rs_w, rs_wlo, has_lo, k0, HH, qbase, t, w, lane, acc);
    if (TAIL > 0)
        decode_chunk16<(TAIL > 0 ? TAIL : 2)>(X, p, rs_z, rs_w, rs_wlo, has_lo, k_full, p.Dv - k_full, qbase, t, w, lane, acc);
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = 64 * w + 32 * ni + 8 * g + 4 * lh;
#pragma unroll
            for (int mi = 0; mi < HMI; ++mi) {
                uint2 o;
                o.x = pack_relu2(acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1]);
                o.y = pack_relu2(acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]);
                *reinterpret_cast<uint2*>(X + (32 * mi + li) * HLD + col) = o;
            }
        }
    for (int l = 0; l < p.n_hidden; ++l) hidden_layer16v2(X, p.frag_hidden[l], p.frag_hidden_lo[l], p.bias_hidden[l], w, lane);
    __syncthreads();
    decode_tail16(X, p, t, qbase);
}

// ---- host side ----------------------------------------------------------------------------------
int pack_fragments_h16(const float* W, int ld, int N, int K, void* P, hipStream_t s, int residual, void* P_lo) {
    const int n_tiles = (N + 31) / 32, nks = (K + 15) / 16;
    const long total = (long)n_tiles * nks * 64;
    int grid = (int)((total + 255) / 256);
    ProfScope prof("pack_fragments" CIAOSR_H16_SUFFIX, s);
    hipLaunchKernelGGL(pack_fragments_h16_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, s, W, ld, N, K,
                       reinterpret_cast<uint4*>(P), n_tiles, nks, residual, reinterpret_cast<uint4*>(P_lo));
    return launch_status("pack_fragments" CIAOSR_H16_SUFFIX);
}

constexpr size_t kFused16Lds = (size_t)HBM_ * HLD * 2 + (size_t)(HBM_ * 4 + 4 * HBM_ + HBM_) * sizeof(float) + (HBM_ + 32 + HBM_) * sizeof(int);

int head_kv_fused_h16(const FusedKVP& p, hipStream_t s) {
    CIAOSR_BIG_LDS(head_kv_fused_h16_kernel, kFused16Lds);
    ProfScope prof("head_kv_fused" CIAOSR_H16_SUFFIX, s);
    hipLaunchKernelGGL(head_kv_fused_h16_kernel, dim3(ceil_div(p.nq, HBM_ / 4)), dim3(256), kFused16Lds, s, p);
    return launch_status("head_kv_fused" CIAOSR_H16_SUFFIX);
}

int head_decode_fused_h16(const FusedQP& p, hipStream_t s) {
#ifdef CIAOSR_DECODE_LDS       // developer experiment (tools/slp_bisect.py): more LDS than the kernel uses = ONE workgroup per CU
    const size_t lds = CIAOSR_DECODE_LDS;
#else
    const size_t lds = (size_t)HBM_ * HLD * 2;
#endif
    CIAOSR_BIG_LDS(head_decode_fused_h16_kernel<0>, lds);
    CIAOSR_BIG_LDS(head_decode_fused_h16_kernel<2>, lds);
    CIAOSR_BIG_LDS(head_decode_fused_h16_kernel<4>, lds);
    CIAOSR_BIG_LDS(head_decode_fused_h16_kernel<8>, lds);
    CIAOSR_BIG_LDS(head_decode_fused_h16_kernel<16>, lds);
    ProfScope prof("head_decode_fused" CIAOSR_H16_SUFFIX, s);
    const dim3 grid(ceil_div(p.nq, HBM_));
    const int tail_steps = ((p.Dv & (HH - 1)) + 15) >> 4;      // k-steps of the ragged last chunk (C = 64: 8, C = 180: 1)
    if (tail_steps == 0)
        hipLaunchKernelGGL(head_decode_fused_h16_kernel<0>, grid, dim3(256), lds, s, p);
    else if (tail_steps <= 2)
        hipLaunchKernelGGL(head_decode_fused_h16_kernel<2>, grid, dim3(256), lds, s, p);
    else if (tail_steps <= 4)
        hipLaunchKernelGGL(head_decode_fused_h16_kernel<4>, grid, dim3(256), lds, s, p);
    else if (tail_steps <= 8)
        hipLaunchKernelGGL(head_decode_fused_h16_kernel<8>, grid, dim3(256), lds, s, p);
    else
        hipLaunchKernelGGL(head_decode_fused_h16_kernel<16>, grid, dim3(256), lds, s, p);
    return launch_status("head_decode_fused" CIAOSR_H16_SUFFIX);
}

}  // namespace CIAOSR_H16_NS
}  // namespace ciaosr

using namespace ciaosr;

// 16-bit fragment packing: out[nt][ks][lane][8 x 16 bit], same byte count for both element types
#if CIAOSR_F16
extern "C" size_t ciaosr_fragment_f16_bytes(int N, int K) { return (size_t)((N + 31) / 32) * ((K + 15) / 16) * 64 * 16; }

extern "C" int ciaosr_pack_fragments_f16(const float* W, int ld, int N, int K, void* out, void* stream) {
    CIAOSR_CHECK_ARG(W && out && N > 0 && K > 0 && ld >= K);
    return f16::pack_fragments_h16(W, ld, N, K, out, (hipStream_t)stream, 0, nullptr);
}

extern "C" int ciaosr_pack_fragments_f16_lo(const float* W, int ld, int N, int K, void* out, void* stream) {
    CIAOSR_CHECK_ARG(W && out && N > 0 && K > 0 && ld >= K);
    return f16::pack_fragments_h16(W, ld, N, K, out, (hipStream_t)stream, 1, nullptr);
}

extern "C" int ciaosr_pack_fragments_f16_pair(const float* W, int ld, int N, int K, void* out_hi, void* out_lo, void* stream) {
    CIAOSR_CHECK_ARG(W && out_hi && out_lo && N > 0 && K > 0 && ld >= K);
    return f16::pack_fragments_h16(W, ld, N, K, out_hi, (hipStream_t)stream, 0, out_lo);
}
#else
#ifdef CIAOSR_PROBE
extern "C" int ciaosr_debug_probe16_read(unsigned long long* host, int n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ciaosr::b16::g_hprobe16), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t ciaosr_fragment_bf16_bytes(int N, int K) { return (size_t)((N + 31) / 32) * ((K + 15) / 16) * 64 * 16; }

extern "C" int ciaosr_pack_fragments_bf16(const float* W, int ld, int N, int K, void* out, void* stream) {
    CIAOSR_CHECK_ARG(W && out && N > 0 && K > 0 && ld >= K);
    return b16::pack_fragments_h16(W, ld, N, K, out, (hipStream_t)stream, 0, nullptr);
}

extern "C" int ciaosr_pack_fragments_bf16_lo(const float* W, int ld, int N, int K, void* out, void* stream) {
    CIAOSR_CHECK_ARG(W && out && N > 0 && K > 0 && ld >= K);
    return b16::pack_fragments_h16(W, ld, N, K, out, (hipStream_t)stream, 1, nullptr);
}

extern "C" int ciaosr_pack_fragments_bf16_pair(const float* W, int ld, int N, int K, void* out_hi, void* out_lo, void* stream) {
    CIAOSR_CHECK_ARG(W && out_hi && out_lo && N > 0 && K > 0 && ld >= K);
    return b16::pack_fragments_h16(W, ld, N, K, out_hi, (hipStream_t)stream, 0, out_lo);
}
#endif
