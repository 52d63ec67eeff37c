// Fused head kernels (hidden width 256, local_size 2): activations never leave the CU.
//
//  head_kv_fused   one workgroup = 32 rows = 8 queries x 4 key samples (64 rows selectable).
//                  index math -> layer-0 rows from the hoisted tables -> phi_k hidden layers
//                  in LDS -> phi_k output layer fused with  logit = sum_d q[d] key[d] w_k[d]  -> softmax
//                  over the 4 samples -> phi_v hidden layers -> phi_v output layer fused with
//                  z = sum_j a_j value_j * w_v,j  -> Z [Q][9C+Cn]                 (ciaosr_net.py:159-216)
//  head_decode_fused   64 queries per workgroup: phi_q on Z (layer 0 streamed through LDS), last
//                  Linear (-> 3) and the bilinear/border residual on the VALU      (ciaosr_net.py:107-108,220-222)
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32).  A operand = the 32 (64) x 256 activation tile in LDS (row stride
// 260 floats: conflict-free ds_read_b128, one float4 feeds 4 MFMAs through the consistent k permutation
// k = 8j + 4h + e); B operand = weights pre-packed on the device into per-wave fragment order
// [n_tile][j][lane][4] so that a wave's fragment is one coalesced 1 KiB load straight from L2 into VGPRs
// (weights are shared by every workgroup and never staged in LDS).  4 waves split the output columns;
// layers run in place: all waves finish reading X, barrier, write bias+ReLU results, barrier.
// Four 32-row workgroups per CU (34.5 KB of LDS each) overlap each other's epilogue/barrier bubbles with MFMAs.

#include "h16_util.h"
#include "index_math.h"
#include "ops.h"

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FH = 256;        // hidden width
constexpr int FLD = FH + 4;    // LDS row stride (floats)
constexpr int FNJ = FH / 8;    // k-chunks of 8 per 256-wide layer

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOobF = 0xFFFFFFF0u;

#ifdef CIAOSR_PROBE      // developer probe build (make probe; tools/head_probe.py): cycle stamps of workgroup phases
__device__ unsigned long long g_hprobe[4096 * 16];
#define HPROBE(slot) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_hprobe[blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define HPROBE(slot) do { } while (0)
#endif

// Epilogue gathers go through buffer descriptors: a column group past the layer width or a missing query row
// gets an out-of-range offset and reads zeros; no branch around the load, so all loads of an epilogue are in
// flight together (per-element conditionals make hipcc wait vmcnt(0) after each one).
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, float4 v) {
    i32x4 iv;
    iv.x = __float_as_int(v.x); iv.y = __float_as_int(v.y); iv.z = __float_as_int(v.z); iv.w = __float_as_int(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(iv, rsrc, (int)byte_off, 0, 0);
}

// ---- fragment packing ---------------------------------------------------------------------------
// W [N][ld] (K valid columns) -> P[nt][j][lane][4]: lane (i = lane&31, h = lane>>5) holds
// W[nt*32 + i][8j + 4h .. 8j + 4h + 3]; rows >= N and columns >= K are zero.
__global__ void pack_fragments_kernel(const float* __restrict__ W, int ld, int N, int K, float* __restrict__ P,
                                      int n_tiles, int nj) {
    const long total = (long)n_tiles * nj * 64;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long t = idx >> 6;
        const int j = (int)(t % nj), nt = (int)(t / nj);
        const int n = nt * 32 + (lane & 31), k = 8 * j + 4 * (lane >> 5);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) {
            const float* r = W + (size_t)n * ld;
            if (k < K) v.x = r[k];
            if (k + 1 < K) v.y = r[k + 1];
            if (k + 2 < K) v.z = r[k + 2];
            if (k + 3 < K) v.w = r[k + 3];
        }
        reinterpret_cast<float4*>(P)[idx] = v;
    }
}

// ---- one MFMA pass: acc[mi][ni] (+)= W_tile . X_tile^T  (SWAPPED operands) --------------------------
// The weight fragment is the MFMA A operand and the activation fragment the B operand, so the result
// tile is D[n][m]: a lane holds ONE activation row m = 32*mi + (lane&31) and the 16 output columns
// n = 32*tile + 8g + 4h + e (g = reg>>2, e = reg&3, h = lane>>5), i.e. four groups of 4 CONSECUTIVE
// columns.  Epilogues are therefore per-row: one key/value row pointer per lane, float4 loads along the
// channel axis, ds_write_b128 / float4 stores, and the reduction over channels is in-register.
// xa: &X[lane row][4h] of this lane; wf: this wave's first n-tile fragment stream (+lane), tile stride in float4
// (templated on the number MT of 32-row MFMA tiles of a workgroup: 1 = 32 rows, 2 = 64 rows)
template <int MT, int NT>
__device__ __forceinline__ void mma_pass(const float* xa, const float4* __restrict__ wf, int nj, long tile_stride,
                                         f32x16 (&acc)[MT][NT]) {
    // software pipeline: weight fragments (L2) and activation fragments (LDS) of step j + 1 are requested before
    // the MFMAs of step j, so neither latency sits between two MFMA groups
    float4 fb[NT], fbn[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) fb[ni] = wf[ni * tile_stride];
    float4 fa[MT], fan[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) fa[mi] = *reinterpret_cast<const float4*>(xa + mi * 32 * FLD);
#pragma unroll 1
    for (int j = 0; j < nj; ++j) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) fan[mi] = fa[mi];
        if (j + 1 < nj) {
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) fbn[ni] = wf[ni * tile_stride + (long)(j + 1) * 64];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) fan[mi] = *reinterpret_cast<const float4*>(xa + mi * 32 * FLD + 8 * (j + 1));
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[ni].x, fa[mi].x, acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[ni].y, fa[mi].y, acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[ni].z, fa[mi].z, acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[ni].w, fa[mi].w, acc[mi][ni], 0, 0, 0);
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) fb[ni] = fbn[ni];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) fa[mi] = fan[mi];
    }
}

// The same pass for a compile-time number NJ of k-chunks, fully unrolled over STATIC register stages (round 3).  The VALU and the
// fp32 MFMA of a gfx950 SIMD share their issue time -- tools/ubench/mfma_valu.hip: every VALU instruction between two
// v_mfma_f32_32x32x2_f32 costs ~4 cycles of matrix-pipe time whichever wave of the SIMD issues it (4 VALU per MFMA: 80 cycles per MFMA
// instead of 64, with one, two or four waves per SIMD) -- and the rolled loop above spends 14 VALU per 8 MFMAs on register rotation
// (fb = fbn, fa = fan) and 64-bit address arithmetic, and waits for the requests of step j + 1 (vmcnt(0): the rotation moves read
// them) before the MFMAs of step j.  Here: weights through a buffer descriptor with a SCALAR offset (lane part in one constant
// VGPR), requested two steps ahead into a ring of three stages, activation fragments one step ahead into two, no moves, counted waits.
//   wrs: descriptor of the fragment array; lane16 = lane * 16; s_off: byte offset of this wave's first fragment (wave-uniform)
__device__ __forceinline__ float4 wload4(__amdgpu_buffer_rsrc_t rsrc, unsigned lane16, unsigned s_off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane16, (int)s_off, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

template <int MT, int NT, int NJ>
__device__ __forceinline__ void mma_pass_u(const float* xa, __amdgpu_buffer_rsrc_t wrs, unsigned lane16, unsigned s_off,
                                           unsigned tile_stride_bytes, f32x16 (&acc)[MT][NT]) {
    static_assert(NJ >= 2, "at least two k-chunks");
    float4 fb[3][NT], fa[2][MT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        fb[0][ni] = wload4(wrs, lane16, s_off + ni * tile_stride_bytes);
        fb[1][ni] = wload4(wrs, lane16, s_off + ni * tile_stride_bytes + 1024u);
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) fa[0][mi] = *reinterpret_cast<const float4*>(xa + mi * 32 * FLD);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j + 2 < NJ) {
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) fb[(j + 2) % 3][ni] = wload4(wrs, lane16, s_off + ni * tile_stride_bytes + (unsigned)(j + 2) * 1024u);
        }
        if (j + 1 < NJ) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) fa[(j + 1) & 1][mi] = *reinterpret_cast<const float4*>(xa + mi * 32 * FLD + 8 * (j + 1));
        }
        // component-major: consecutive MFMAs go to different accumulators wherever the pass has more than one
#define CIAOSR_MMA_C(c)                                                                                                          \
        _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                                                                        \
            _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                                                                    \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j % 3][ni].c, fa[j & 1][mi].c, acc[mi][ni], 0, 0, 0);
        CIAOSR_MMA_C(x) CIAOSR_MMA_C(y) CIAOSR_MMA_C(z) CIAOSR_MMA_C(w)
#undef CIAOSR_MMA_C
        __builtin_amdgcn_sched_barrier(0);       // the requests stay at the top of their step
    }
}

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t frag_rsrc(const void* frag) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(frag), 0, 0xFFFFFFFFu, 0x00020000);
}

template <int MT, int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
}

// The bias of a layer is its accumulators' INITIAL value (round 4; the 16-bit kernels did this since round 2): the 64 moves that cleared the
// accumulators load the bias instead and the layer's epilogue is one v_max_f32 per value -- every VALU instruction of these kernels costs the
// shared SIMD ~5 matrix-pipe cycles.  bias + sum of products instead of sum of products + bias: the last bit may differ.
template <int MT>
__device__ __forceinline__ void bias_acc(f32x16 (&acc)[MT][2], const float* __restrict__ bias, int w, int lh) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = *reinterpret_cast<const float4*>(bias + 64 * w + 32 * ni + 8 * g + 4 * lh);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                acc[mi][ni][4 * g] = b.x; acc[mi][ni][4 * g + 1] = b.y; acc[mi][ni][4 * g + 2] = b.z; acc[mi][ni][4 * g + 3] = b.w;
            }
        }
}

template <int MT>
__device__ __forceinline__ void store_relu_tile(float* X, const f32x16 (&acc)[MT][2], int w, int li, int lh) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = 64 * w + 32 * ni + 8 * g + 4 * lh;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                float4 o;
                o.x = fmaxf(acc[mi][ni][4 * g], 0.f);
                o.y = fmaxf(acc[mi][ni][4 * g + 1], 0.f);
                o.z = fmaxf(acc[mi][ni][4 * g + 2], 0.f);
                o.w = fmaxf(acc[mi][ni][4 * g + 3], 0.f);
                *reinterpret_cast<float4*>(X + (32 * mi + li) * FLD + col) = o;
            }
        }
}

template <int MT>
__device__ __forceinline__ void hidden_layer(float* X, const void* __restrict__ frag, const float* __restrict__ bias,
                                             int w, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[MT][2];
    bias_acc<MT>(acc, bias, w, lh);
    mma_pass_u<MT, 2, FNJ>(X + li * FLD + 4 * lh, frag_rsrc(frag), (unsigned)lane * 16u, (unsigned)(2 * w) * (FNJ * 1024u), FNJ * 1024u, acc);
    __syncthreads();   // every wave has finished reading X
    store_relu_tile<MT>(X, acc, w, li, lh);
    __syncthreads();
}


template <int MT>
__device__ __forceinline__ void build_rows(float* X, const FusedChain& c, const int* s_kpix, const float* s_t4, int t) {
    // 64 rows x 64 float4: thread handles float4 column (t & 63) of rows (t >> 6) + 4*s
    const int n4 = t & 63;
    float4 tw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) tw[e] = *reinterpret_cast<const float4*>(c.tail + (size_t)(4 * n4 + e) * c.ld_tail);
    // the four coordinate terms as PACKED fp32 FMAs on column pairs (round 4): 8 + 4 instead of 16 + 4 VALU instructions per float4, the same
    // FMA chain per element (tv + w_ry ry, + w_rx rx, + w_sy sy, + w_sx sx); broadcast / same-lane source selections only (Makefile)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 wy01 = {tw[0].x, tw[1].x}, wy23 = {tw[2].x, tw[3].x}, wx01 = {tw[0].y, tw[1].y}, wx23 = {tw[2].y, tw[3].y};
    const f32x2 vy01 = {tw[0].z, tw[1].z}, vy23 = {tw[2].z, tw[3].z}, vx01 = {tw[0].w, tw[1].w}, vx23 = {tw[2].w, tw[3].w};
    for (int r = t >> 6; r < 32 * MT; r += 4) {
        const float4 tv = reinterpret_cast<const float4*>(c.table + (size_t)s_kpix[r] * FH)[n4];
        const float4 t4 = *reinterpret_cast<const float4*>(s_t4 + 4 * r);        // rel_y rel_x scale_y scale_x
        const f32x2 ry = {t4.x, t4.x}, rx = {t4.y, t4.y}, sy = {t4.z, t4.z}, sx = {t4.w, t4.w};
        f32x2 a = __builtin_elementwise_fma(wy01, ry, f32x2{tv.x, tv.y}), b = __builtin_elementwise_fma(wy23, ry, f32x2{tv.z, tv.w});
        a = __builtin_elementwise_fma(wx01, rx, a); b = __builtin_elementwise_fma(wx23, rx, b);
        a = __builtin_elementwise_fma(vy01, sy, a); b = __builtin_elementwise_fma(vy23, sy, b);
        a = __builtin_elementwise_fma(vx01, sx, a); b = __builtin_elementwise_fma(vx23, sx, b);
        *reinterpret_cast<float4*>(X + r * FLD + 4 * n4) = make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(b.x, 0.f), fmaxf(b.y, 0.f));
    }
}


// 32-row workgroups (MT = 1, 8 queries) are the default: four fit a CU (34.5 KB of LDS each) and overlap each other's
// non-MFMA phases -- measured 1.46 -> 1.41 ms at C2, 1.96 -> 1.68 ms at C5 (C = 180) and 22.5 -> 21.8 ms at the 192 tile against
// 64-row workgroups (two per CU), although every weight fragment then serves half as many rows.
template <int MT>      // 32-row MFMA tiles per workgroup: 1 (32 rows = 8 queries) or 2 (64 rows)
__global__ __launch_bounds__(256, 2) void head_kv_fused_kernel(FusedKVP p) {
    constexpr int BM = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                                   // [64][260]
    float* s_t4 = X + BM * FLD;                       // [64][4]  rel_y rel_x scale_y scale_x
    float* s_part = s_t4 + BM * 4;                    // [4][64]  per-wave partial logits
    float* s_attn = s_part + 4 * BM;                  // [64]
    int* s_kpix = reinterpret_cast<int*>(s_attn + BM);  // [64]
    int* s_qpix = s_kpix + BM;                        // [16]
    int* s_goff = s_qpix + BM / 4;                    // [64]  logit-table row of each (query, sample) row

    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int qbase = blockIdx.x * (BM / 4);          // local query index of row 0
    HPROBE(0);

    // ---- index math: one thread per row (ciaosr_net.py:145-193) ---------------------------------
    int bad = 0;
    if (t < BM) {
        const int ql = qbase + (t >> 2), j = t & 3;
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
            if (j == 0) s_qpix[t >> 2] = qin ? iy * p.W + ix : -1;
            if (qin) {
                const int oy = s.ky - iy, ox = s.kx - ix;      // key pixel relative to the query pixel
                if (oy >= -1 && oy <= 1 && ox >= -1 && ox <= 1) goff = (iy * p.W + ix) * 9 + (oy + 1) * 3 + (ox + 1);
                else bad = 1;                                   // exotic cell: not a 3x3 neighbour -> MFMA path
            }
        } else if (j == 0) {
            s_qpix[t >> 2] = -1;
        }
        s_kpix[t] = kpix;
        s_goff[t] = goff;
#pragma unroll
        for (int e = 0; e < 4; ++e) s_t4[4 * t + e] = t4[e];
    }
    const bool table = p.G != nullptr && !__syncthreads_or(bad);   // (also the barrier after the index phase)
    if (p.G == nullptr) __syncthreads();

    // ================= phi_k =====================================================================
    HPROBE(1);
    build_rows<MT>(X, p.k, s_kpix, s_t4, t);
    __syncthreads();
    HPROBE(2);
    for (int l = 0; l < p.k.n_hidden; ++l) {
        hidden_layer<MT>(X, p.k.frag_hidden[l], p.k.bias_hidden[l], w, lane);
        if (l < 3) HPROBE(10 + l);
    }
    HPROBE(3);

    if (table) {
        // logit = h4 . G[query pixel, key offset] + c  (exact fold of the output layer, head_ops.hip qk_rows):
        // 4 threads per row, float4-interleaved over the 256 hidden units
        const bool rv = (t >> 2) < BM;                    // 32-row variant: the upper half of the threads idles
        const int row = rv ? (t >> 2) : 0, part = t & 3;
        const int go = rv ? s_goff[row] : -1;
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, p.g_bytes, 0x00020000);
        const unsigned gbase = go >= 0 ? (unsigned)go * (unsigned)p.ldg * 4u : kOobF;
        float4 gv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) gv[i] = bload4(rs_g, gbase == kOobF ? kOobF : gbase + (unsigned)(16 * i + 4 * part) * 4u);
        const float cterm = (go >= 0 && part == 0) ? p.G[(size_t)go * p.ldg + 256] : 0.f;
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 x = *reinterpret_cast<const float4*>(X + row * FLD + 16 * i + 4 * part);
            a += x.x * gv[i].x + x.y * gv[i].y + x.z * gv[i].z + x.w * gv[i].w;
        }
        a += cterm;
        a += quad_xor1(a);
        a += quad_xor2(a);
        if (part == 0 && rv) {
            s_part[row] = a;
            s_part[BM + row] = 0.f;
            s_part[2 * BM + row] = 0.f;
            s_part[3 * BM + row] = 0.f;
        }
    } else
    // output layer fused with the logit dot product: wave w takes 32-column units w, w+4, ...
    {
        float part[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) part[mi] = 0.f;
        const int n_units = (p.k.n_out + 31) >> 5;
        const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U), 0, p.u_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_bk =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.k.bias_out), 0, (unsigned)p.k.n_out * 4u, 0x00020000);
        unsigned koff[MT], qoff[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int m = 32 * mi + li;
            koff[mi] = (unsigned)s_kpix[m] * (unsigned)p.ldu * 4u;
            const int qp = s_qpix[m >> 2];
            qoff[mi] = qp >= 0 ? (unsigned)qp * (unsigned)p.ldu * 4u : kOobF;     // missing query row reads zeros
        }
        for (int u = w; u < n_units; u += 4) {
            f32x16 acc[MT][1];
            zero_acc<MT, 1>(acc);
            mma_pass_u<MT, 1, FNJ>(X + li * FLD + 4 * lh, frag_rsrc(p.k.frag_out), (unsigned)lane * 16u, (unsigned)u * (FNJ * 1024u), 0u, acc);
            // logit += sum_d q[d] * (key[d] * (w_k[d] + b[d]))   (ciaosr_net.py:203,214)
            float4 bv[4], kv[MT][4], qv[MT][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * u + 8 * g + 4 * lh;
                const unsigned doff = d0 < p.k.n_out ? (unsigned)d0 * 4u : kOobF;
                bv[g] = bload4(rs_bk, doff);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    kv[mi][g] = bload4(rs_u, doff == kOobF ? kOobF : koff[mi] + doff);
                    qv[mi][g] = bload4(rs_u, (doff == kOobF || qoff[mi] == kOobF) ? kOobF : qoff[mi] + doff);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    part[mi] += qv[mi][g].x * (kv[mi][g].x * (acc[mi][0][4 * g] + bv[g].x)) +
                                qv[mi][g].y * (kv[mi][g].y * (acc[mi][0][4 * g + 1] + bv[g].y)) +
                                qv[mi][g].z * (kv[mi][g].z * (acc[mi][0][4 * g + 2] + bv[g].z)) +
                                qv[mi][g].w * (kv[mi][g].w * (acc[mi][0][4 * g + 3] + bv[g].w));
        }
        // the two half-waves hold different channels of the same rows
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            part[mi] += __shfl_xor(part[mi], 32, 64);
            if (lh == 0) s_part[w * BM + 32 * mi + li] = part[mi];
        }
    }
    __syncthreads();
    // softmax over the 4 key samples of each query (ciaosr_net.py:214-215)
    if (t < BM / 4) {
        float lg[4], m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 4 * t + j;
            lg[j] = (s_part[row] + s_part[BM + row] + s_part[2 * BM + row] + s_part[3 * BM + row]) / p.softmax_scale;
            m = fmaxf(m, lg[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { lg[j] = expf(lg[j] - m); den += lg[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) s_attn[4 * t + j] = lg[j] / den;
    }
    // (the barrier inside build_rows' caller below also orders s_attn)

    // ================= phi_v =====================================================================
    HPROBE(4);
    build_rows<MT>(X, p.v, s_kpix, s_t4, t);   // all waves are past their last read of X (barrier above)
    __syncthreads();
    HPROBE(5);
    for (int l = 0; l < p.v.n_hidden; ++l) {
        hidden_layer<MT>(X, p.v.frag_hidden[l], p.v.bias_hidden[l], w, lane);
        if (l < 3) HPROBE(13 + l);
    }
    HPROBE(6);
    {
        const int n_units = (p.v.n_out + 31) >> 5;
        const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U), 0, p.u_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_bv =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.v.bias_out), 0, (unsigned)p.v.n_out * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_z =
            __builtin_amdgcn_make_buffer_rsrc(p.Z, 0, (unsigned)((size_t)p.nq * p.ldz * 4), 0x00020000);
        unsigned voff[MT], zoff[MT];
        float av[MT];
        const int jsel = li & 3;     // this lane's key sample; also the channel group it stores
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int m = 32 * mi + li;
            voff[mi] = (unsigned)s_kpix[m] * (unsigned)p.ldu * 4u;
            av[mi] = s_attn[m];
            const int ql = qbase + (m >> 2);
            zoff[mi] = ql < p.nq ? (unsigned)ql * (unsigned)p.ldz * 4u : kOobF;
        }
        for (int u = w; u < n_units; u += 4) {
            // the epilogue's gathers (bias, value rows) do not depend on the MFMA pass: request them first so their L2
            // latency hides behind the 256 MFMAs of this unit
            float4 bv[4], vv[MT][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * u + 8 * g + 4 * lh;
                const unsigned doff = d0 < p.v.n_out ? (unsigned)d0 * 4u : kOobF;
                bv[g] = bload4(rs_bv, doff);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) vv[mi][g] = bload4(rs_u, doff == kOobF ? kOobF : voff[mi] + doff);
            }
            f32x16 acc[MT][1];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {      // the output layer's bias as the accumulators' initial value
                    acc[mi][0][4 * g] = bv[g].x; acc[mi][0][4 * g + 1] = bv[g].y; acc[mi][0][4 * g + 2] = bv[g].z; acc[mi][0][4 * g + 3] = bv[g].w;
                }
            mma_pass_u<MT, 1, FNJ>(X + li * FLD + 4 * lh, frag_rsrc(p.v.frag_out), (unsigned)lane * 16u, (unsigned)u * (FNJ * 1024u), 0u, acc);
            // z[d] = sum_j a_j * (value_j[d] * (w_v,j[d] + b[d]))   (ciaosr_net.py:206,215): the 4 samples of a
            // query sit in 4 adjacent lanes -> quad reduction, then lane j stores channel group j as one float4
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                float4 zsel = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 z;
                    z.x = av[mi] * (vv[mi][g].x * acc[mi][0][4 * g]);
                    z.y = av[mi] * (vv[mi][g].y * acc[mi][0][4 * g + 1]);
                    z.z = av[mi] * (vv[mi][g].z * acc[mi][0][4 * g + 2]);
                    z.w = av[mi] * (vv[mi][g].w * acc[mi][0][4 * g + 3]);
                    z.x += quad_xor1(z.x); z.y += quad_xor1(z.y); z.z += quad_xor1(z.z); z.w += quad_xor1(z.w);
                    z.x += quad_xor2(z.x); z.y += quad_xor2(z.y); z.z += quad_xor2(z.z); z.w += quad_xor2(z.w);
                    if (jsel == g) zsel = z;
                }
                const int d0 = 32 * u + 8 * jsel + 4 * lh;
                bstore4(rs_z, (zoff[mi] == kOobF || d0 >= p.v.n_out) ? kOobF : zoff[mi] + (unsigned)d0 * 4u, zsel);
            }
        }
    }
    HPROBE(7);
#ifdef CIAOSR_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_hprobe[blockIdx.x * 16 + 8] = __builtin_amdgcn_s_getreg(63492);    // HW_REG_HW_ID
        g_hprobe[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_getreg(63508);    // HW_REG_XCC_ID
    }
#endif
}

// ---------------------------------------------------------------------------------------------
template <int MT>      // 32-query MFMA tiles per workgroup: 2 (64 queries) or 1 (the tail launch)
__global__ __launch_bounds__(256, 2) void head_decode_fused_kernel(FusedQP p) {
    constexpr int BM = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;   // [64][260]
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int qbase = blockIdx.x * BM;

    // layer 0: K = Dv streamed through X in chunks of 256 columns
    const __amdgpu_buffer_rsrc_t rs_zin =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Z), 0, (unsigned)((size_t)p.nq * p.ldz * 4), 0x00020000);
    f32x16 acc[MT][2];
    bias_acc<MT>(acc, p.bias_in, w, lh);
    for (int k0 = 0; k0 < p.Dv; k0 += FH) {
        const int kc = min(FH, p.Dv - k0);          // multiple of 8
        if (k0 > 0) __syncthreads();                 // previous chunk fully consumed
        {   // stage the 64 x 256 chunk of Z: 16 float4 per thread, all requested before the first LDS store, through a
            // buffer descriptor (a guarded load inside a load -> store loop costs one HBM round trip per iteration)
            const int c4 = (t & 63) * 4;
            float4 zv[BM / 4];
#pragma unroll
            for (int i = 0; i < BM / 4; ++i) {
                const int ql = qbase + (t >> 6) + 4 * i;
                zv[i] = bload4(rs_zin, (ql < p.nq && c4 < kc) ? ((unsigned)ql * (unsigned)p.ldz + (unsigned)(k0 + c4)) * 4u : kOobF);
            }
#pragma unroll
            for (int i = 0; i < BM / 4; ++i) *reinterpret_cast<float4*>(X + ((t >> 6) + 4 * i) * FLD + c4) = zv[i];
        }
        __syncthreads();
        const unsigned s_in = ((unsigned)(2 * w) * (unsigned)p.nj_in + (unsigned)(k0 >> 3)) * 1024u;
        if (kc == FH)
            mma_pass_u<MT, 2, FNJ>(X + li * FLD + 4 * lh, frag_rsrc(p.frag_in), (unsigned)lane * 16u, s_in, (unsigned)p.nj_in * 1024u, acc);
        else if (kc == FH / 2)
            mma_pass_u<MT, 2, FNJ / 2>(X + li * FLD + 4 * lh, frag_rsrc(p.frag_in), (unsigned)lane * 16u, s_in, (unsigned)p.nj_in * 1024u, acc);
        else
            mma_pass<MT, 2>(X + li * FLD + 4 * lh,
                        reinterpret_cast<const float4*>(p.frag_in) + ((size_t)(2 * w) * p.nj_in + (k0 >> 3)) * 64 + lane,
                        kc >> 3, (long)p.nj_in * 64, acc);
    }
    __syncthreads();
    store_relu_tile<MT>(X, acc, w, li, lh);
    __syncthreads();
    for (int l = 0; l < p.n_hidden; ++l) hidden_layer<MT>(X, p.frag_hidden[l], p.bias_hidden[l], w, lane);

    // last Linear (256 -> 3): 4 threads per row, 64 columns each
    const bool rv = (t >> 2) < BM;                        // 32-query variant: the upper half of the threads idles
    const int row = rv ? (t >> 2) : 0, part = t & 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    {
        const float* xr = X + row * FLD + 64 * part;
        const float* w0 = p.w_last + 64 * part;
        const float* w1 = w0 + p.ld_last;
        const float* w2 = w1 + p.ld_last;
#pragma unroll 4
        for (int n = 0; n < 64; n += 4) {
            const float4 x = *reinterpret_cast<const float4*>(xr + n);
            const float4 u0 = *reinterpret_cast<const float4*>(w0 + n);
            const float4 u1 = *reinterpret_cast<const float4*>(w1 + n);
            const float4 u2 = *reinterpret_cast<const float4*>(w2 + n);
            a0 += x.x * u0.x + x.y * u0.y + x.z * u0.z + x.w * u0.w;
            a1 += x.x * u1.x + x.y * u1.y + x.z * u1.z + x.w * u1.w;
            a2 += x.x * u2.x + x.y * u2.y + x.z * u2.z + x.w * u2.w;
        }
    }
    a0 += quad_xor1(a0); a0 += quad_xor2(a0);
    a1 += quad_xor1(a1); a1 += quad_xor2(a1);
    a2 += quad_xor1(a2); a2 += quad_xor2(a2);
    const int ql = qbase + row;
    if (part < 3 && rv && ql < p.nq) {
        const long q = p.q0 + ql;
        float v = (part == 0 ? a0 : part == 1 ? a1 : a2) + p.b_last[part];
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
            const float* img = p.x_lr + (size_t)part * p.H * p.W;
            const bool y1ok = y0 + 1 < p.H, x1ok = x0 + 1 < p.W;
            float r = img[(size_t)y0 * p.W + x0] * (wx0 * wy0);
            if (x1ok) r += img[(size_t)y0 * p.W + x0 + 1] * (wx1 * wy0);
            if (y1ok) r += img[(size_t)(y0 + 1) * p.W + x0] * (wx0 * wy1);
            if (y1ok && x1ok) r += img[(size_t)(y0 + 1) * p.W + x0 + 1] * (wx1 * wy1);
            v += r;
        }
        p.rgb[q * 3 + part] = v;
    }
}

// ---- host side ----------------------------------------------------------------------------------
size_t fragment_floats(int N, int K) { return (size_t)((N + 31) / 32) * ((K + 7) / 8) * 64 * 4; }

int pack_fragments(const float* W, int ld, int N, int K, float* P, hipStream_t s) {
    const int n_tiles = (N + 31) / 32, nj = (K + 7) / 8;
    const long total = (long)n_tiles * nj * 64;
    int grid = (int)((total + 255) / 256);
    ProfScope prof("pack_fragments", s);
    hipLaunchKernelGGL(pack_fragments_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, s, W, ld, N, K, P, n_tiles, nj);
    return launch_status("pack_fragments");
}

template <int MT>
constexpr size_t fused_kv_lds() {
    return (size_t)(32 * MT * FLD + 32 * MT * 4 + 4 * 32 * MT + 32 * MT) * sizeof(float) + (32 * MT + 8 * MT + 32 * MT) * sizeof(int);
}

int head_kv_fused(const FusedKVP& p, hipStream_t s) {
    CIAOSR_BIG_LDS(head_kv_fused_kernel<1>, fused_kv_lds<1>());
    CIAOSR_BIG_LDS(head_kv_fused_kernel<2>, fused_kv_lds<2>());
    // 64-row workgroups (two per CU) once they fill the chip a few times over: every weight fragment feeds two row tiles, and since
    // the k loop stopped waiting per step (mma_pass_u) that outweighs the fourth resident workgroup of the 32-row form
    // (C3 tile: 19.9 -> 18.9 ms; before the unrolled pass the 64-row form measured 3-14 % slower).  Same result bit for bit.
    const int rows = p.rows_per_wg == 64 || (p.rows_per_wg != 32 && ceil_div(p.nq, 16) >= 2048) ? 64 : 32;
    ProfScope prof("head_kv_fused", s);
    if (rows == 64)
        hipLaunchKernelGGL(head_kv_fused_kernel<2>, dim3(ceil_div(p.nq, 16)), dim3(256), fused_kv_lds<2>(), s, p);
    else
        hipLaunchKernelGGL(head_kv_fused_kernel<1>, dim3(ceil_div(p.nq, 8)), dim3(256), fused_kv_lds<1>(), s, p);
    return launch_status("head_kv_fused");
}

int head_decode_fused(const FusedQP& p, hipStream_t s) {
    CIAOSR_BIG_LDS(head_decode_fused_kernel<2>, (size_t)64 * FLD * sizeof(float));
    CIAOSR_BIG_LDS(head_decode_fused_kernel<1>, (size_t)32 * FLD * sizeof(float));
    // 32-query workgroups by default (four per CU; measured better than 64-query ones at C2 and at the 192 tile: 0.36 -> 0.29 ms,
    // 3.93 -> 3.83 ms), like the phi_k/phi_v kernel
    const int rows = p.rows_per_wg == 64 ? 64 : 32;   // 64: experiments (ciaosr_options_t.decode_rows)
    ProfScope prof("head_decode_fused", s);
    if (rows == 64)
        hipLaunchKernelGGL(head_decode_fused_kernel<2>, dim3(ceil_div(p.nq, 64)), dim3(256), (size_t)64 * FLD * sizeof(float), s, p);
    else
        hipLaunchKernelGGL(head_decode_fused_kernel<1>, dim3(ceil_div(p.nq, 32)), dim3(256), (size_t)32 * FLD * sizeof(float), s, p);
    return launch_status("head_decode_fused");
}

}  // namespace ciaosr

using namespace ciaosr;

extern "C" size_t ciaosr_fragment_floats(int N, int K) { return fragment_floats(N, K); }

extern "C" int ciaosr_pack_fragments_f32(const float* W, int ld, int N, int K, float* out, void* stream) {
    CIAOSR_CHECK_ARG(W && out && N > 0 && K > 0 && ld >= K);
    return pack_fragments(W, ld, N, K, out, (hipStream_t)stream);
}

#ifdef CIAOSR_PROBE
extern "C" int ciaosr_debug_probe_read(unsigned long long* host, int n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ciaosr::g_hprobe), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
#endif
