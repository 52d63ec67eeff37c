// Staged head kernels of LocalImplicitSRNet.query_rgb (ciaosr_net.py:113-224).
//
//  head_indices     nearest LR index of each query and of its J key samples          (:145,159-183)
//  head_rows        layer-1 activations of imnet_k / imnet_v per (query, shift) row via the exact
//                   layer-1 hoist: h1 = relu(T[key pixel] + W1[:, tail] . [rel, scale]) where
//                   T = U . W1[:, :D]^T + b1 is one GEMM per LR tile     (:185-202)
//  local_attention  K4: 4 logits per query, softmax, weighted value sum               (:203-216)
//  decode_residual  last Linear of imnet_q (-> 3) + bilinear/border LR residual       (:107-108,221)
//
// One wavefront (64 lanes) per row / query; lanes stride the channel dimension with float4.
#include "h16_util.h"
#include "common.h"
#include "index_math.h"
#include "ops.h"

namespace ciaosr {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ long cell0_index(long q, int chunk) { return chunk > 0 ? (q / chunk) * chunk : 0; }

// ---------------------------------------------------------------------------------------------
__global__ void head_indices_kernel(const float* __restrict__ coord, const float* __restrict__ cell, long q0,
                                    int nq, int chunk, int H, int W, int local_size, int J,
                                    int* __restrict__ q_idx, int* __restrict__ k_idx, float* __restrict__ rel) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const long q = q0 + i;
    const float cy = coord[2 * q], cx = coord[2 * q + 1];
    const long c0 = cell0_index(q, chunk);
    const float c0y = cell[2 * c0], c0x = cell[2 * c0 + 1];
    const int iy = nearest_index(cy, H), ix = nearest_index(cx, W);
    q_idx[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? iy * W + ix : -1;
    for (int j = 0; j < J; ++j) {
        const KeySample s = key_sample(cy, cx, c0y, c0x, H, W, j, local_size);
        k_idx[(size_t)i * J + j] = s.ky * W + s.kx;
        if (rel) {
            rel[((size_t)i * J + j) * 2] = s.rel_y;
            rel[((size_t)i * J + j) * 2 + 1] = s.rel_x;
        }
    }
}

// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void head_rows_kernel(HeadRowsP p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)p.nq * p.J) return;
    const int i = (int)(row / p.J), j = (int)(row - (long)i * p.J);
    const long q = p.q0 + i;
    const float cy = p.coord[2 * q], cx = p.coord[2 * q + 1];
    const long c0 = cell0_index(q, p.chunk);
    const KeySample s = key_sample(cy, cx, p.cell[2 * c0], p.cell[2 * c0 + 1], p.H, p.W, j, p.local_size);
    const int kpix = s.ky * p.W + s.kx;
    const float sy = mul_rn(p.cell[2 * q], (float)p.H);       // scale_ = cell * [H, W]  (:191-193)
    const float sx = mul_rn(p.cell[2 * q + 1], (float)p.W);
    if (lane == 0) {
        p.k_idx[row] = kpix;
        if (j == 0) {
            const int iy = nearest_index(cy, p.H), ix = nearest_index(cx, p.W);
            p.q_idx[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? iy * p.W + ix : -1;
        }
    }
    {
        const float4* T = reinterpret_cast<const float4*>(p.Tk + (size_t)kpix * p.wk0);
        const float* tw = p.tailK;
        float4* o = reinterpret_cast<float4*>(p.Hk + (size_t)row * p.wk0);
        for (int n4 = lane; n4 < (p.wk0 >> 2); n4 += 64) {
            const float4 t = T[n4];
            float r[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 w = *reinterpret_cast<const float4*>(tw + (size_t)(4 * n4 + e) * p.ld_tail_k);
                float v = r[e] + w.x * s.rel_y + w.y * s.rel_x + w.z * sy + w.w * sx;
                r[e] = p.relu_k == CIAOSR_ACT_RELU ? fmaxf(v, 0.f) : p.relu_k == CIAOSR_ACT_SIN ? sinf(v) : p.relu_k == CIAOSR_ACT_COS ? cosf(v) : v;
            }
            o[n4] = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
    {
        const float4* T = reinterpret_cast<const float4*>(p.Tv + (size_t)kpix * p.wv0);
        const float* tw = p.tailV;
        float4* o = reinterpret_cast<float4*>(p.Hv + (size_t)row * p.wv0);
        for (int n4 = lane; n4 < (p.wv0 >> 2); n4 += 64) {
            const float4 t = T[n4];
            float r[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 w = *reinterpret_cast<const float4*>(tw + (size_t)(4 * n4 + e) * p.ld_tail_v);
                float v = r[e] + w.x * s.rel_y + w.y * s.rel_x + w.z * sy + w.w * sx;
                r[e] = p.relu_v == CIAOSR_ACT_RELU ? fmaxf(v, 0.f) : p.relu_v == CIAOSR_ACT_SIN ? sinf(v) : p.relu_v == CIAOSR_ACT_COS ? cosf(v) : v;
            }
            o[n4] = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
}

// The same rows with ONE WAVE PER QUERY (round 6; 256-wide first layers = one float4 per lane): the J (query, sample) rows of a query share its
// coordinate loads and index math, the lane's eight tail-weight rows are fetched once instead of J times, and -- the point -- all 2 J table
// rows are requested before the first store.  The one-row-per-wave form above puts 2 KB of stores behind a three-deep dependent load chain
// (coord -> key sample -> table row) with one row in flight per wave: 0.33 of the HBM peak on its nominal bytes, latency- not bandwidth-bound
// (profiles/r6_c3tile_staged_rooflines.txt).  Same arithmetic per element: bitwise the same rows.
template <int J>
__global__ __launch_bounds__(256) void head_rows_query_kernel(HeadRowsP p) {
    const int lane = threadIdx.x & 63;
    const int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (i >= p.nq) return;
    const long q = p.q0 + i;
    const float cy = p.coord[2 * q], cx = p.coord[2 * q + 1];
    const long c0 = cell0_index(q, p.chunk);
    const float c0y = p.cell[2 * c0], c0x = p.cell[2 * c0 + 1];
    const float sy = mul_rn(p.cell[2 * q], (float)p.H);       // scale_ = cell * [H, W]  (:191-193)
    const float sx = mul_rn(p.cell[2 * q + 1], (float)p.W);
    KeySample s[J];
    int kpix[J];
    float4 tk[J], tv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        s[j] = key_sample(cy, cx, c0y, c0x, p.H, p.W, j, p.local_size);
        kpix[j] = s[j].ky * p.W + s[j].kx;
        tk[j] = reinterpret_cast<const float4*>(p.Tk + (size_t)kpix[j] * p.wk0)[lane];
        tv[j] = reinterpret_cast<const float4*>(p.Tv + (size_t)kpix[j] * p.wv0)[lane];
    }
    float4 wkt[4], wvt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        wkt[e] = *reinterpret_cast<const float4*>(p.tailK + (size_t)(4 * lane + e) * p.ld_tail_k);
        wvt[e] = *reinterpret_cast<const float4*>(p.tailV + (size_t)(4 * lane + e) * p.ld_tail_v);
    }
    if (lane == 0) {
        const int iy = nearest_index(cy, p.H), ix = nearest_index(cx, p.W);
        p.q_idx[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? iy * p.W + ix : -1;
    }
    auto act = [](float v, int code) -> float {
        return code == CIAOSR_ACT_RELU ? fmaxf(v, 0.f) : code == CIAOSR_ACT_SIN ? sinf(v) : code == CIAOSR_ACT_COS ? cosf(v) : v;
    };
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const long row = (long)i * J + j;
        if (lane == 0) p.k_idx[row] = kpix[j];
        float rk[4] = {tk[j].x, tk[j].y, tk[j].z, tk[j].w}, rv[4] = {tv[j].x, tv[j].y, tv[j].z, tv[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            rk[e] = act(rk[e] + wkt[e].x * s[j].rel_y + wkt[e].y * s[j].rel_x + wkt[e].z * sy + wkt[e].w * sx, p.relu_k);
            rv[e] = act(rv[e] + wvt[e].x * s[j].rel_y + wvt[e].y * s[j].rel_x + wvt[e].z * sy + wvt[e].w * sx, p.relu_v);
        }
        reinterpret_cast<float4*>(p.Hk + (size_t)row * p.wk0)[lane] = make_float4(rk[0], rk[1], rk[2], rk[3]);
        reinterpret_cast<float4*>(p.Hv + (size_t)row * p.wv0)[lane] = make_float4(rv[0], rv[1], rv[2], rv[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// K1 gather rows: the MLP inputs as the reference assembles them (net:176-196), one wavefront per (query, sample).
// HBM traffic per query (fp32, C=64): write 4*580*4 + 4*644*4 + 576*4 B (21 936 B incl. coords).
// ---------------------------------------------------------------------------------------------
struct GatherRowsP {
    const float* U; int ldu, D, Dv;
    const float* coord; const float* cell;
    int Q, chunk, H, W, local_size, J;
    float* q_rows; int ldq;
    float* inp_k; int ldk;
    float* inp_v; int ldv;
    int* q_idx; int* k_idx;
};

__global__ __launch_bounds__(256) void gather_rows_kernel(GatherRowsP p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)p.Q * p.J) return;
    const long q = row / p.J;
    const int j = (int)(row - q * p.J);
    const float cy = p.coord[2 * q], cx = p.coord[2 * q + 1];
    const long c0 = cell0_index(q, p.chunk);
    const KeySample s = key_sample(cy, cx, p.cell[2 * c0], p.cell[2 * c0 + 1], p.H, p.W, j, p.local_size);
    const int kpix = s.ky * p.W + s.kx;
    const float4* src = reinterpret_cast<const float4*>(p.U + (size_t)kpix * p.ldu);
    float4* ok = reinterpret_cast<float4*>(p.inp_k + (size_t)row * p.ldk);
    float4* ov = reinterpret_cast<float4*>(p.inp_v + (size_t)row * p.ldv);
    for (int t = lane; t < (p.Dv >> 2); t += 64) {
        const float4 v = src[t];
        ov[t] = v;
        if (t < (p.D >> 2)) ok[t] = v;
    }
    if (lane == 0) {
        const float4 tail = make_float4(s.rel_y, s.rel_x, mul_rn(p.cell[2 * q], (float)p.H), mul_rn(p.cell[2 * q + 1], (float)p.W));
        ok[p.D >> 2] = tail;
        ov[p.Dv >> 2] = tail;
        p.k_idx[row] = kpix;
    }
    if (j == 0) {
        const int iy = nearest_index(cy, p.H), ix = nearest_index(cx, p.W);
        const bool inside = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        if (lane == 0) p.q_idx[q] = inside ? iy * p.W + ix : -1;
        float4* oq = reinterpret_cast<float4*>(p.q_rows + (size_t)q * p.ldq);
        const float4* qs = reinterpret_cast<const float4*>(p.U + (size_t)(inside ? iy * p.W + ix : 0) * p.ldu);
        for (int t = lane; t < (p.D >> 2); t += 64) oq[t] = inside ? qs[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ---------------------------------------------------------------------------------------------
// K4 local attention.  HBM traffic per query (fp32, C=64): wk 4*576*4 + wv 4*640*4 + z 640*4 B.
// ---------------------------------------------------------------------------------------------


template <int J>
__global__ __launch_bounds__(256) void local_attention_kernel(LocalAttnP p) {
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= p.Q) return;
    const int qi = p.q_idx[q];
    int kp[J];
#pragma unroll
    for (int j = 0; j < J; ++j) kp[j] = p.k_idx[q * J + j];
    float logit[J];
#pragma unroll
    for (int j = 0; j < J; ++j) logit[j] = 0.f;
    if (qi >= 0) {
        const float4* qrow = reinterpret_cast<const float4*>(p.U + (size_t)qi * p.ldu);
        for (int t = lane; t < (p.D >> 2); t += 64) {
            const float4 qv = qrow[t];
            _Pragma("unroll") for (int j = 0; j < J; ++j) {
                const float4 kv = reinterpret_cast<const float4*>(p.U + (size_t)kp[j] * p.ldu)[t];
                const float4 wv = reinterpret_cast<const float4*>(p.wk + (size_t)(q * J + j) * p.ldwk)[t];
                logit[j] += qv.x * (kv.x * wv.x) + qv.y * (kv.y * wv.y) + qv.z * (kv.z * wv.z) + qv.w * (kv.w * wv.w);
            }
        }
    }
    float m = -INFINITY;
    _Pragma("unroll") for (int j = 0; j < J; ++j) {
        logit[j] = wsum(logit[j]) / p.scale;
        m = fmaxf(m, logit[j]);
    }
    float den = 0.f;
    _Pragma("unroll") for (int j = 0; j < J; ++j) {
        logit[j] = expf(logit[j] - m);
        den += logit[j];
    }
    _Pragma("unroll") for (int j = 0; j < J; ++j) logit[j] /= den;
    float4* zo = reinterpret_cast<float4*>(p.z + (size_t)q * p.ldz);
    for (int t = lane; t < (p.Dv >> 2); t += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        _Pragma("unroll") for (int j = 0; j < J; ++j) {
            const float4 vv = reinterpret_cast<const float4*>(p.U + (size_t)kp[j] * p.ldu)[t];
            const float4 ww = reinterpret_cast<const float4*>(p.wv + (size_t)(q * J + j) * p.ldwv)[t];
            const float a = logit[j];
            acc.x += a * (vv.x * ww.x);
            acc.y += a * (vv.y * ww.y);
            acc.z += a * (vv.z * ww.z);
            acc.w += a * (vv.w * ww.w);
        }
        zo[t] = acc;
    }
}

// K4 with 16-bit wk / wv / z (11 056 B per query at C = 64 against 22 064 in fp32): the staged route's HBM-bound kernel with
// the three big operands in bf16 or IEEE half -- same fp32 arithmetic on the widened values, z rounded to nearest even (half: saturating).
// One wave per query as above; a lane moves 8 B (4 elements) of wk / wv / z per step.
struct LocalAttn16P {
    const float* U;
    int ldu, D, Dv;
    const int* q_idx;
    const int* k_idx;
    const unsigned short* wk; int ldwk;
    const unsigned short* wv; int ldwv;
    unsigned short* z; int ldz;
    int Q, J;
    float scale;
};

template <bool F16>
__device__ __forceinline__ float4 widen4(uint2 v) {
    return make_float4(h16_lo<F16>(v.x), h16_hi<F16>(v.x), h16_lo<F16>(v.y), h16_hi<F16>(v.y));
}

// VW = elements a lane moves per step: 8 (16 B of wk / wv / z, two float4 of U) where every row length and leading dimension is a multiple
// of 8 -- at 4 the kernel issues as many memory instructions as the fp32 one for half the bytes and runs no faster -- else 4.
template <int J, bool F16, int VW>
__global__ __launch_bounds__(256) void local_attention_h16_kernel(LocalAttn16P p) {
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= p.Q) return;
    const int qi = p.q_idx[q];
    int kp[J];
#pragma unroll
    for (int j = 0; j < J; ++j) kp[j] = p.k_idx[q * J + j];
    float logit[J];
#pragma unroll
    for (int j = 0; j < J; ++j) logit[j] = 0.f;
    constexpr int NV = VW / 4;                               // float4 per step
    if (qi >= 0) {
        const float4* qrow = reinterpret_cast<const float4*>(p.U + (size_t)qi * p.ldu);
        for (int t = lane; t < p.D / VW; t += 64) {
            float4 qv[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) qv[v] = qrow[NV * t + v];
            _Pragma("unroll") for (int j = 0; j < J; ++j) {
                const float4* krow = reinterpret_cast<const float4*>(p.U + (size_t)kp[j] * p.ldu);
                const unsigned short* wrow = p.wk + (size_t)(q * J + j) * p.ldwk;
                float4 wv[NV];
                if constexpr (VW == 8) {
                    const uint4 w8 = reinterpret_cast<const uint4*>(wrow)[t];
                    wv[0] = widen4<F16>(make_uint2(w8.x, w8.y)); wv[1] = widen4<F16>(make_uint2(w8.z, w8.w));
                } else {
                    wv[0] = widen4<F16>(reinterpret_cast<const uint2*>(wrow)[t]);
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float4 kv = krow[NV * t + v];
                    logit[j] += qv[v].x * (kv.x * wv[v].x) + qv[v].y * (kv.y * wv[v].y) + qv[v].z * (kv.z * wv[v].z) + qv[v].w * (kv.w * wv[v].w);
                }
            }
        }
    }
    float m = -INFINITY;
    _Pragma("unroll") for (int j = 0; j < J; ++j) {
        logit[j] = wsum(logit[j]) / p.scale;
        m = fmaxf(m, logit[j]);
    }
    float den = 0.f;
    _Pragma("unroll") for (int j = 0; j < J; ++j) {
        logit[j] = expf(logit[j] - m);
        den += logit[j];
    }
    _Pragma("unroll") for (int j = 0; j < J; ++j) logit[j] /= den;
    unsigned short* zrow = p.z + (size_t)q * p.ldz;
    for (int t = lane; t < p.Dv / VW; t += 64) {
        float4 acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        _Pragma("unroll") for (int j = 0; j < J; ++j) {
            const float4* vrow = reinterpret_cast<const float4*>(p.U + (size_t)kp[j] * p.ldu);
            const unsigned short* wrow = p.wv + (size_t)(q * J + j) * p.ldwv;
            float4 ww[NV];
            if constexpr (VW == 8) {
                const uint4 w8 = reinterpret_cast<const uint4*>(wrow)[t];
                ww[0] = widen4<F16>(make_uint2(w8.x, w8.y)); ww[1] = widen4<F16>(make_uint2(w8.z, w8.w));
            } else {
                ww[0] = widen4<F16>(reinterpret_cast<const uint2*>(wrow)[t]);
            }
            const float a = logit[j];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float4 vv = vrow[NV * t + v];
                acc[v].x += a * (vv.x * ww[v].x);
                acc[v].y += a * (vv.y * ww[v].y);
                acc[v].z += a * (vv.z * ww[v].z);
                acc[v].w += a * (vv.w * ww[v].w);
            }
        }
        if constexpr (VW == 8) {
            const uint2 lo = pack_h16x4<F16>(acc[0].x, acc[0].y, acc[0].z, acc[0].w), hi = pack_h16x4<F16>(acc[1].x, acc[1].y, acc[1].z, acc[1].w);
            reinterpret_cast<uint4*>(zrow)[t] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
            reinterpret_cast<uint2*>(zrow)[t] = pack_h16x4<F16>(acc[0].x, acc[0].y, acc[0].z, acc[0].w);
        }
    }
}

template <bool F16, int VW>
static int local_attention_h16_vw(const LocalAttn16P& p, hipStream_t s) {
    const dim3 grid(ceil_div(p.Q, 4));
    if (p.J == 4) hipLaunchKernelGGL((local_attention_h16_kernel<4, F16, VW>), grid, dim3(256), 0, s, p);
    else if (p.J == 9) hipLaunchKernelGGL((local_attention_h16_kernel<9, F16, VW>), grid, dim3(256), 0, s, p);
    else if (p.J == 1) hipLaunchKernelGGL((local_attention_h16_kernel<1, F16, VW>), grid, dim3(256), 0, s, p);
    else return CIAOSR_ERR_BAD_ARG;
    return CIAOSR_OK;
}

template <bool F16>
static int local_attention_h16(const LocalAttn16P& p, hipStream_t s) {
    ProfScope prof(F16 ? "local_attention_f16" : "local_attention_bf16", s);
    const bool wide = ((p.D | p.Dv | p.ldwk | p.ldwv | p.ldz | p.ldu) & 7) == 0 && (((size_t)p.wk | (size_t)p.wv | (size_t)p.z) & 15) == 0;
    const int rc = wide ? local_attention_h16_vw<F16, 8>(p, s) : local_attention_h16_vw<F16, 4>(p, s);
    if (rc != CIAOSR_OK) return rc;
    return launch_status(F16 ? "local_attention_f16" : "local_attention_bf16");
}

static int local_attention_16_entry(bool f16, const float* unfold, int ld_u, int C, int Cn, const int* q_idx, const int* k_idx, const void* wk,
                                    int ld_wk, const void* wv, int ld_wv, void* z, int ld_z, int Q, int J, float softmax_scale, void* stream) {
    CIAOSR_CHECK_ARG(unfold && q_idx && k_idx && wk && wv && z && Q > 0);
    CIAOSR_CHECK_ARG((J == 1 || J == 4 || J == 9) && (C & 3) == 0 && (Cn & 3) == 0);
    CIAOSR_CHECK_ARG((ld_u & 3) == 0 && (ld_wk & 3) == 0 && (ld_wv & 3) == 0 && (ld_z & 3) == 0);
    CIAOSR_CHECK_ARG(((size_t)wk & 7) == 0 && ((size_t)wv & 7) == 0 && ((size_t)z & 7) == 0);
    LocalAttn16P p{unfold, ld_u, 9 * C, 9 * C + Cn, q_idx, k_idx, reinterpret_cast<const unsigned short*>(wk), ld_wk,
                   reinterpret_cast<const unsigned short*>(wv), ld_wv, reinterpret_cast<unsigned short*>(z), ld_z, Q, J, softmax_scale};
    return f16 ? local_attention_h16<true>(p, (hipStream_t)stream) : local_attention_h16<false>(p, (hipStream_t)stream);
}

extern "C" int ciaosr_local_attention_bf16(const float* unfold, int ld_u, int C, int Cn, const int* q_idx, const int* k_idx, const void* wk,
                                           int ld_wk, const void* wv, int ld_wv, void* z, int ld_z, int Q, int J, float softmax_scale,
                                           void* stream) {
    return local_attention_16_entry(false, unfold, ld_u, C, Cn, q_idx, k_idx, wk, ld_wk, wv, ld_wv, z, ld_z, Q, J, softmax_scale, stream);
}
extern "C" int ciaosr_local_attention_f16(const float* unfold, int ld_u, int C, int Cn, const int* q_idx, const int* k_idx, const void* wk,
                                          int ld_wk, const void* wv, int ld_wv, void* z, int ld_z, int Q, int J, float softmax_scale,
                                          void* stream) {
    return local_attention_16_entry(true, unfold, ld_u, C, Cn, q_idx, k_idx, wk, ld_wk, wv, ld_wv, z, ld_z, Q, J, softmax_scale, stream);
}

// ---------------------------------------------------------------------------------------------
// decode: rgb = W_last . h + b_last + bilinear_border(x_lr; coord)
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void decode_residual_kernel(DecodeP p) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= p.nq) return;
    const long q = p.q0 + i;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const float4* hr = reinterpret_cast<const float4*>(p.h + (size_t)i * p.ldh);
    const float4* w0 = reinterpret_cast<const float4*>(p.w);
    const float4* w1 = reinterpret_cast<const float4*>(p.w + p.ldw);
    const float4* w2 = reinterpret_cast<const float4*>(p.w + 2 * p.ldw);
    for (int t = lane; t < (p.width >> 2); t += 64) {
        const float4 h = hr[t];
        const float4 x = w0[t], y = w1[t], z = w2[t];
        a0 += h.x * x.x + h.y * x.y + h.z * x.z + h.w * x.w;
        a1 += h.x * y.x + h.y * y.y + h.z * y.z + h.w * y.w;
        a2 += h.x * z.x + h.y * z.y + h.z * z.z + h.w * z.w;
    }
    a0 = wsum(a0); a1 = wsum(a1); a2 = wsum(a2);
    if (lane < 3) {
        float v = (lane == 0 ? a0 : lane == 1 ? a1 : a2) + p.b[lane];
        if (p.x_lr) {
            // F.grid_sample(bilinear, padding_mode='border', align_corners=False)  (ciaosr_net.py:107-108)
            const float cy = p.coord[2 * q], cx = p.coord[2 * q + 1];
            float fy = sub_rn(mul_rn(add_rn(cy, 1.0f), (float)p.H * 0.5f), 0.5f);
            float fx = sub_rn(mul_rn(add_rn(cx, 1.0f), (float)p.W * 0.5f), 0.5f);
            fy = fminf((float)(p.H - 1), fmaxf(fy, 0.f));
            fx = fminf((float)(p.W - 1), fmaxf(fx, 0.f));
            const float y0f = floorf(fy), x0f = floorf(fx);
            const int y0 = (int)y0f, x0 = (int)x0f;
            const float wy1 = fy - y0f, wy0 = (y0f + 1.f) - fy;
            const float wx1 = fx - x0f, wx0 = (x0f + 1.f) - fx;
            const float* img = p.x_lr + (size_t)lane * p.H * p.W;
            const bool y1ok = y0 + 1 < p.H, x1ok = x0 + 1 < p.W;
            float r = img[(size_t)y0 * p.W + x0] * (wx0 * wy0);
            if (x1ok) r += img[(size_t)y0 * p.W + x0 + 1] * (wx1 * wy0);
            if (y1ok) r += img[(size_t)(y0 + 1) * p.W + x0] * (wx0 * wy1);
            if (y1ok && x1ok) r += img[(size_t)(y0 + 1) * p.W + x0 + 1] * (wx1 * wy1);
            v += r;
        }
        p.rgb[q * 3 + lane] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Rows of the logit-table GEMM (exact output-layer fold of imnet_k, SURVEY B.2 "analogous"):
//   logit = sum_d q[d] key[d] (sum_n W5[d][n] h[n] + b5[d]) = sum_n h[n] G[n] + c,
//   G = W5^T (q*key), c = b5 . (q*key), and (q*key) depends only on (query pixel, key pixel) with the key
//   pixel one of the 3x3 neighbours of the query pixel: 9 rows per LR pixel instead of one 576-wide output
//   layer per (query, sample) row.  Row r = p*9 + (oy+1)*3 + (ox+1): A[r][d] = U[p][d] * U[p+o][d].
// ---------------------------------------------------------------------------------------------
template <int OUT>        // OUT: 0 = fp32 rows; 1 / 2 = rows written as bf16 / half (operand of the 16-bit GEMM); 3 = no rows, the bias term c only
__global__ __launch_bounds__(256) void qk_rows_kernel(const float* __restrict__ U, int ldu, int D, int H, int W, long row0,
                                                      int nrows, const float* __restrict__ b5, float* __restrict__ A,
                                                      float* __restrict__ G, int ldg) {
    constexpr bool B16 = OUT == 1 || OUT == 2;
    const int lane = threadIdx.x & 63;
    const long rl = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rl >= nrows) return;
    const long r = row0 + rl;
    const int pix = (int)(r / 9), o = (int)(r - (long)pix * 9);
    const int y = pix / W, x = pix - y * W;
    const int ky = y + o / 3 - 1, kx = x + o % 3 - 1;
    float4* a = reinterpret_cast<float4*>(A + (size_t)rl * D);
    uint2* a16 = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(A) + (size_t)rl * D);
    float c = 0.f;
    if (ky < 0 || ky >= H || kx < 0 || kx >= W) {          // this (query pixel, key pixel) pair cannot occur
        for (int t = lane; OUT != 3 && t < (D >> 2); t += 64) {
            if (B16) a16[t] = make_uint2(0u, 0u);
            else a[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        const float4* q = reinterpret_cast<const float4*>(U + (size_t)pix * ldu);
        const float4* k = reinterpret_cast<const float4*>(U + ((size_t)ky * W + kx) * ldu);
        const float4* b = reinterpret_cast<const float4*>(b5);
        for (int t = lane; t < (D >> 2); t += 64) {
            const float4 qv = q[t], kv = k[t], bv = b[t];
            const float4 v = make_float4(qv.x * kv.x, qv.y * kv.y, qv.z * kv.z, qv.w * kv.w);
            if (OUT == 3) {}
            else if (B16) a16[t] = pack_h16x4<OUT == 2>(v.x, v.y, v.z, v.w);
            else a[t] = v;
            c += v.x * bv.x + v.y * bv.y + v.z * bv.z + v.w * bv.w;
        }
        c = wsum(c);
    }
    if (lane == 0) G[(size_t)r * ldg + 256] = c;
}

// The same table as nine 3x3 CONVOLUTIONS (Winograd route, dense_wino_f32.hip wino_table_f32): with U[p][c*9 + k] = F[p + k][c] (the
// unfold, zero padded), G[p, o][n] = sum_k sum_c Pi_o[p + k][c] W5[c*9 + k][n], Pi_o[x][c] = F[x][c] F[x + o][c] -- a 64 -> 256
// convolution of the product map Pi_o, which a 576-deep GEMM row per (p, o) recomputes nine times over.  Maps: Pi[o][pix][C].
__global__ __launch_bounds__(256) void qk_maps_kernel(const float* __restrict__ F, int ldf, int C4, int H, int W, float* __restrict__ Pi) {
    const long HW = (long)H * W, per = HW * C4;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= 9 * per) return;
    const int o = (int)(i / per);
    const long r = i - o * per;
    const int pix = (int)(r / C4), c4 = (int)(r - (long)pix * C4);
    const int y = pix / W, x = pix - y * W;
    const int ky = y + o / 3 - 1, kx = x + o % 3 - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ky >= 0 && ky < H && kx >= 0 && kx < W) {
        const float4 a = reinterpret_cast<const float4*>(F + (size_t)pix * ldf)[c4];
        const float4 b = reinterpret_cast<const float4*>(F + ((size_t)ky * W + kx) * ldf)[c4];
        v = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
    }
    reinterpret_cast<float4*>(Pi)[i] = v;
}

// ---------------------------------------------------------------------------------------------
// make_coord + cell of an Ht x Wt target grid on the device (ciaosr.py:237-243, mmedit make_coord)
// ---------------------------------------------------------------------------------------------
__global__ void make_coord_cell_kernel(float* __restrict__ coord, float* __restrict__ cell, int Ht, int Wt) {
    const long n = (long)Ht * Wt;
    const float cy = (float)(2.0 / (double)Ht), cx = (float)(2.0 / (double)Wt);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / Wt), x = (int)(i - (long)y * Wt);
        reinterpret_cast<float2*>(coord)[i] = make_float2(pixel_centre(y, Ht), pixel_centre(x, Wt));
        reinterpret_cast<float2*>(cell)[i] = make_float2(cy, cx);
    }
}

// ---- host wrappers --------------------------------------------------------------------------
int head_indices(const float* coord, const float* cell, long q0, int nq, int chunk, int H, int W, int local_size,
                 int* q_idx, int* k_idx, float* rel, hipStream_t s) {
    const int J = local_size == 1 ? 1 : (local_size == 2 ? 4 : 9);
    ProfScope prof("head_indices", s);
    hipLaunchKernelGGL(head_indices_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, s, coord, cell, q0, nq, chunk, H,
                       W, local_size, J, q_idx, k_idx, rel);
    return launch_status("head_indices");
}

int head_rows(const HeadRowsP& p, hipStream_t s) {
    ProfScope prof("head_rows", s);
    if (p.wk0 == 256 && p.wv0 == 256 && p.J == 4)       // the configs' head: one wave per query, every table row of the query in flight at once
        hipLaunchKernelGGL(head_rows_query_kernel<4>, dim3(ceil_div((long)p.nq, 4)), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(head_rows_kernel, dim3(ceil_div((long)p.nq * p.J, 4)), dim3(256), 0, s, p);
    return launch_status("head_rows");
}

int qk_rows(const float* U, int ldu, int D, int H, int W, long row0, int nrows, const float* bias_out, float* A, float* G,
            int ldg, int rows_h16, hipStream_t s) {
    ProfScope prof("head_qk_rows", s);
    if (rows_h16 == 2)
        hipLaunchKernelGGL(qk_rows_kernel<2>, dim3(ceil_div(nrows, 4)), dim3(256), 0, s, U, ldu, D, H, W, row0, nrows, bias_out,
                           A, G, ldg);
    else if (rows_h16 == 3)
        hipLaunchKernelGGL(qk_rows_kernel<3>, dim3(ceil_div(nrows, 4)), dim3(256), 0, s, U, ldu, D, H, W, row0, nrows, bias_out,
                           A, G, ldg);
    else if (rows_h16 == 1)
        hipLaunchKernelGGL(qk_rows_kernel<1>, dim3(ceil_div(nrows, 4)), dim3(256), 0, s, U, ldu, D, H, W, row0, nrows, bias_out,
                           A, G, ldg);
    else
        hipLaunchKernelGGL(qk_rows_kernel<0>, dim3(ceil_div(nrows, 4)), dim3(256), 0, s, U, ldu, D, H, W, row0, nrows, bias_out,
                           A, G, ldg);
    return launch_status("qk_rows");
}

int qk_maps(const float* F, int ldf, int C, int H, int W, float* Pi, hipStream_t s) {
    CIAOSR_CHECK_ARG(F && Pi && (C & 3) == 0 && (ldf & 3) == 0 && aligned16(F) && aligned16(Pi));
    ProfScope prof("head_qk_maps", s);
    const long n = 9L * H * W * (C / 4);
    hipLaunchKernelGGL(qk_maps_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, F, ldf, C / 4, H, W, Pi);
    return launch_status("qk_maps");
}

// out[n][k] (bf16 or half, row stride K) = W[k][n] (fp32, row stride ld): the [N][K] operand of the 16-bit NT GEMM
template <bool F16>
__global__ void transpose_cast_h16_kernel(const float* __restrict__ W, int ld, int K, int N, unsigned short* __restrict__ out) {
    const long n_el = (long)K * N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_el; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K), n = (int)(i / K);
        out[i] = to_h16<F16>(W[(size_t)k * ld + n]);
    }
}

int transpose_cast_h16(const float* W, int ld, int K, int N, unsigned short* out, bool f16, hipStream_t s) {
    ProfScope prof("transpose_cast_h16", s);
    const long n_el = (long)K * N;
    if (f16)
        hipLaunchKernelGGL(transpose_cast_h16_kernel<true>, dim3((int)((n_el + 255) / 256)), dim3(256), 0, s, W, ld, K, N, out);
    else
        hipLaunchKernelGGL(transpose_cast_h16_kernel<false>, dim3((int)((n_el + 255) / 256)), dim3(256), 0, s, W, ld, K, N, out);
    return launch_status("transpose_cast_h16");
}

int local_attention(const LocalAttnP& p, hipStream_t s) {
    ProfScope prof("local_attention", s);
    if (p.J == 4)
        hipLaunchKernelGGL(local_attention_kernel<4>, dim3(ceil_div(p.Q, 4)), dim3(256), 0, s, p);
    else if (p.J == 9)
        hipLaunchKernelGGL(local_attention_kernel<9>, dim3(ceil_div(p.Q, 4)), dim3(256), 0, s, p);
    else if (p.J == 1)
        hipLaunchKernelGGL(local_attention_kernel<1>, dim3(ceil_div(p.Q, 4)), dim3(256), 0, s, p);
    else
        return CIAOSR_ERR_UNSUPPORTED;
    return launch_status("local_attention");
}

int decode_residual(const DecodeP& p, hipStream_t s) {
    ProfScope prof("decode_residual", s);
    hipLaunchKernelGGL(decode_residual_kernel, dim3(ceil_div(p.nq, 4)), dim3(256), 0, s, p);
    return launch_status("decode_residual");
}

}  // namespace ciaosr

using namespace ciaosr;

extern "C" int ciaosr_head_indices_f32(const float* coord, const float* cell, int Q, int chunk, int H, int W,
                                       int local_size, int* q_idx, int* k_idx, float* rel, void* stream) {
    CIAOSR_CHECK_ARG(coord && cell && q_idx && k_idx && Q > 0 && H > 0 && W > 0);
    CIAOSR_CHECK_ARG(local_size >= 1 && local_size <= 3);
    return head_indices(coord, cell, 0, Q, chunk, H, W, local_size, q_idx, k_idx, rel, (hipStream_t)stream);
}

extern "C" int ciaosr_make_coord_cell_f32(float* coord, float* cell, int Ht, int Wt, void* stream) {
    CIAOSR_CHECK_ARG(coord && cell && Ht > 0 && Wt > 0);
    ProfScope prof("make_coord_cell", (hipStream_t)stream);
    long n = (long)Ht * Wt;
    int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL(make_coord_cell_kernel, dim3(grid > 2048 ? 2048 : grid), dim3(256), 0, (hipStream_t)stream, coord,
                       cell, Ht, Wt);
    return launch_status("make_coord_cell");
}

extern "C" int ciaosr_local_attention_f32(const float* unfold, int ld_u, int C, int Cn, const int* q_idx,
                                          const int* k_idx, const float* wk, int ld_wk, const float* wv,
                                          int ld_wv, float* z, int ld_z, int Q, int J, float softmax_scale,
                                          void* stream) {
    CIAOSR_CHECK_ARG(unfold && q_idx && k_idx && wk && wv && z && Q > 0);
    CIAOSR_CHECK_ARG((J == 1 || J == 4 || J == 9) && (C & 3) == 0 && (Cn & 3) == 0);
    CIAOSR_CHECK_ARG((ld_u & 3) == 0 && (ld_wk & 3) == 0 && (ld_wv & 3) == 0 && (ld_z & 3) == 0);
    LocalAttnP p{unfold, ld_u, 9 * C, 9 * C + Cn, q_idx, k_idx, wk, ld_wk, wv, ld_wv, z, ld_z, Q, J,
                 softmax_scale};
    return local_attention(p, (hipStream_t)stream);
}

extern "C" int ciaosr_gather_rows_f32(const float* unfold, int ld_u, int C, int Cn, const float* coord, const float* cell,
                                      int Q, int chunk, int H, int W, int local_size, float* q_rows, int ld_q,
                                      float* inp_k, int ld_k, float* inp_v, int ld_v, int* q_idx, int* k_idx,
                                      void* stream) {
    CIAOSR_CHECK_ARG(unfold && coord && cell && q_rows && inp_k && inp_v && q_idx && k_idx && Q > 0 && H > 0 && W > 0);
    CIAOSR_CHECK_ARG(local_size >= 1 && local_size <= 3 && (C & 3) == 0 && (Cn & 3) == 0 && C > 0 && Cn >= 0);
    const int D = 9 * C, Dv = D + Cn;
    CIAOSR_CHECK_ARG((ld_u & 3) == 0 && (ld_q & 3) == 0 && (ld_k & 3) == 0 && (ld_v & 3) == 0);
    CIAOSR_CHECK_ARG(ld_u >= Dv && ld_q >= D && ld_k >= D + 4 && ld_v >= Dv + 4);
    const int J = local_size == 1 ? 1 : (local_size == 2 ? 4 : 9);
    GatherRowsP p{unfold, ld_u, D, Dv, coord, cell, Q, chunk, H, W, local_size, J, q_rows, ld_q, inp_k, ld_k, inp_v, ld_v,
                  q_idx, k_idx};
    ProfScope prof("gather_rows", (hipStream_t)stream);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div((long)Q * J, 4)), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("gather_rows");
}

extern "C" int ciaosr_decode_residual_f32(const float* h, int ld_h, int width, const float* w_last, int ld_w,
                                          const float* b_last, const float* x_lr_nchw, const float* coord, int Q, int H,
                                          int W, float* rgb, void* stream) {
    CIAOSR_CHECK_ARG(h && w_last && b_last && coord && rgb && Q > 0 && H > 0 && W > 0);
    CIAOSR_CHECK_ARG(width > 0 && (width & 3) == 0 && (ld_h & 3) == 0 && (ld_w & 3) == 0 && ld_h >= width && ld_w >= width);
    DecodeP p{h, ld_h, width, w_last, ld_w, b_last, x_lr_nchw, coord, 0, Q, H, W, rgb};
    return decode_residual(p, (hipStream_t)stream);
}
