// Exact-fp32 MFMA GEMM for gfx950:  C = act((A . B^T + bias) * alpha)
//
// v_mfma_f32_32x32x2_f32 (64 cycles/SIMD, 157 TFLOP/s chip peak, bitwise an fmaf chain).
// Tile 128x128x16 (BK below), 256 threads = 4 waves as 2(M) x 2(N), each wave 64x64 = 2x2 MFMA tiles
// (64 accumulator VGPRs).  A and B tiles are staged through LDS with a register prefetch of the
// next K-tile and two LDS buffers (one barrier per K-tile).  LDS rows are padded to BK + 4 floats so
// the per-lane float4 fragment reads (ds_read_b128) are bank-conflict free; one float4 feeds four
// MFMAs because the k index inside a fragment is free to be permuted consistently on A and B:
// lane (i = l&31, h = l>>5) holds k = 8j + 4h + e for MFMA e of chunk j.
//
// Replaces torch addmm / conv2d(1x1) / conv_transpose2d-as-GEMM of the reference
// (mlp_refiner.py:79-89, arch_csnln.py:452-453,475,499-500,511,516).
#include <type_traits>

#include "common.h"

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef CIAOSR_GEMM32_TNK
#define CIAOSR_GEMM32_TNK 24
#endif
// k-tile depth 16 (round 3): 40 KB of LDS per workgroup = THREE workgroups per CU (768 slots: the 2304 tiles of the attn.V contraction
// are 3.0 rounds instead of 4.5 of 512), affordable since the k loop carries no per-tile VALU work any more.  C3 tile: csa_attn_v
// 6.12 -> 5.73 ms, head_table 0.41 -> 0.28 ms (32 stays selectable at build time)
#ifndef CIAOSR_GEMM32_BK
#define CIAOSR_GEMM32_BK 16
#endif
constexpr int BM = 128, BN = 128, BK = CIAOSR_GEMM32_BK;   // BK = 32 or 16
constexpr int F4R = BK / 4;              // float4 per tile row
constexpr int NSA = BM * F4R / 256;      // tile float4 per thread (A, and B of the NT form)
constexpr int NSB = BK * (BN / 4) / 256; // ... of the [k][n] image of B
constexpr int LDS_A = BK + 4;   // 36 floats / row (NT layouts)
constexpr int LDS_BKN = BN + 4; // 132 floats / row for the [k][n] image
constexpr int A_TILE = BM * LDS_A;                                        // 4608 floats
constexpr int B_TILE = (BN * LDS_A > BK * LDS_BKN) ? BN * LDS_A : BK * LDS_BKN;  // 4608

struct GemmP {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K, lda, ldb, ldc;
    float alpha, slope;
    int act;
    int tiles_n, n_wg;
    unsigned a_bytes, b_bytes;   // buffer-descriptor extents (< 4 GiB)
    int splitk, kt_per_split;    // skinny problems: the K loop is split over blockIdx.y into partial slabs
    float* partial;              // [splitk][M][N]
    const float2* a_stats;       // optional: A holds logits; a_stats[row * a_stats_stride] = (row max x log2 e, 1 / sum of exp): the staging
    int a_stats_stride;          // writes exp2(a log2 e - max log2 e) / sum into LDS (row softmax applied on the fly)
    int tn_per_wg, groups_n;     // short K: a workgroup walks tn_per_wg consecutive column tiles of one row tile as ONE software
                                 // pipeline (the first k-tile of the next column tile is prefetched under the last of the current)
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOobG = 0xFFFFFFF0u;

// Tile loads go through buffer descriptors: rows past M / N and k past K get an out-of-range offset and the
// hardware returns zeros without a branch ("cond ? load : 0" makes hipcc branch around every load and wait
// vmcnt(0) per element, turning the 8 loads of a K-tile into dependent round trips).  A float4 that straddles
// K (K % 4 != 0) is masked component-wise when it is written to LDS, i.e. after the MFMA phase.
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
// the same load with a wave-uniform byte offset in the scalar operand (not part of the range check: an out-of-range per-lane offset
// stays out of range)
__device__ __forceinline__ float4 buf_ld4s(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, unsigned s_off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, (int)s_off, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ float4 mask4(float4 v, int first, int limit) {
    if (first + 1 >= limit) v.y = 0.f;
    if (first + 2 >= limit) v.z = 0.f;
    if (first + 3 >= limit) v.w = 0.f;
    return v;
}

// MULTI: several column tiles per workgroup in one pipeline; plain epilogue only ((acc + bias) * alpha, no activation, no split-K)
// SM: A holds logits and p.a_stats their row statistics -- the staging writes probabilities (row softmax applied on the fly)
template <bool B_KN, bool MULTI, bool SM = false>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][A_TILE]
    float* Bs = smem + 2 * A_TILE;     // [2][B_TILE]

    // XCD-aware bijective remap: each XCD (bid % 8) walks a contiguous run of tiles, and
    // consecutive tiles share the A row-block, so A panels are re-read from that XCD's L2.
    const int bid = blockIdx.x;
    const int q8 = p.n_wg >> 3, r8 = p.n_wg & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int m0 = (lid / p.groups_n) * BM;
    const int tile0 = (lid % p.groups_n) * p.tn_per_wg;
    const int ntl = min(p.tn_per_wg, p.tiles_n - tile0);          // column tiles of this workgroup

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int li = lane & 31, lh = lane >> 5;

    float4 ra[4], rb[4];
    const int nk_total = (p.K + BK - 1) / BK;
    const int kt0 = blockIdx.y * p.kt_per_split;
    const int kt1 = min(nk_total, kt0 + p.kt_per_split);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);

    // The k loop carries no per-load address or validity arithmetic on the VALU (every VALU instruction of a SIMD costs its fp32 matrix
    // pipe ~4 cycles, tools/ubench/mfma_valu.hip; ablation at the attn.V shape: MFMAs + fragment reads alone 0.92 of the peak, the full
    // loop 0.79, of which ~80 VALU per 64 MFMAs of offsets, selects and masks were the larger part): per-thread byte offsets are formed
    // once per column tile (a row / column past the edge = an out-of-range offset: the descriptor returns zeros), the k-tile enters as
    // the SCALAR offset of the load, and the selects of a ragged K run on a uniform branch that only the last k-tile takes.
    unsigned offA[NSA], offB[NSA > NSB ? NSA : NSB];
#pragma unroll
    for (int s = 0; s < NSA; ++s) {
        const int idx = t + 256 * s;
        const int gm = m0 + idx / F4R;
        offA[s] = gm < p.M ? ((unsigned)gm * (unsigned)p.lda + (unsigned)((idx % F4R) * 4)) * 4u : kOobG;
    }
    auto set_cols = [&](int n0) __attribute__((always_inline)) {
        if (!B_KN) {
#pragma unroll
            for (int s = 0; s < NSA; ++s) {
                const int idx = t + 256 * s;
                const int gn = n0 + idx / F4R;
                offB[s] = gn < p.N ? ((unsigned)gn * (unsigned)p.ldb + (unsigned)((idx % F4R) * 4)) * 4u : kOobG;
            }
        } else {
#pragma unroll
            for (int s = 0; s < NSB; ++s) {
                const int idx = t + 256 * s;
                const int gn = n0 + (idx & 31) * 4;
                offB[s] = gn < p.N ? ((unsigned)(idx >> 5) * (unsigned)p.ldb + (unsigned)gn) * 4u : kOobG;      // rows k >= K: masked in load_tiles
            }
        }
    };
    const bool k_ragged = (p.K % BK) != 0;                       // uniform
    int cols_n0 = -1;
    auto load_tiles = [&](int kt, int n0) __attribute__((always_inline)) {
        if (n0 != cols_n0) { set_cols(n0); cols_n0 = n0; }       // uniform: once per column tile
        const int k0 = kt * BK;
        const unsigned sk = (unsigned)k0 * 4u;
        if (k_ragged && k0 + BK > p.K) {                         // last k-tile of a ragged K: float4s wholly past K are not read
#pragma unroll
            for (int s = 0; s < NSA; ++s) {
                const bool in = k0 + ((t + 256 * s) % F4R) * 4 < p.K;
                ra[s] = buf_ld4s(rs_a, in ? offA[s] : kOobG, sk);
                if (!B_KN) rb[s] = buf_ld4s(rs_b, in ? offB[s] : kOobG, sk);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NSA; ++s) ra[s] = buf_ld4s(rs_a, offA[s], sk);
            if (!B_KN) {
#pragma unroll
                for (int s = 0; s < NSA; ++s) rb[s] = buf_ld4s(rs_b, offB[s], sk);
            }
        }
        if (B_KN) {
            // the k-tile sits in the SCALAR offset, which the descriptor's range check does not see: rows k >= K of a ragged last k-tile
            // would read whatever follows B (a NaN there times the zero-masked A columns is NaN), so they take the out-of-range per-lane
            // offset on the same uniform branch as the ragged A columns
            const unsigned skb = (unsigned)k0 * (unsigned)p.ldb * 4u;
            if (k_ragged && k0 + BK > p.K) {
#pragma unroll
                for (int s = 0; s < NSB; ++s) rb[s] = buf_ld4s(rs_b, k0 + ((t + 256 * s) >> 5) < p.K ? offB[s] : kOobG, skb);
            } else {
#pragma unroll
                for (int s = 0; s < NSB; ++s) rb[s] = buf_ld4s(rs_b, offB[s], skb);
            }
        }
    };
    // row-softmax statistics of this thread's four A rows (rows t / 8 + 32 s), when A holds logits
    float2 ast[4];
    if constexpr (SM) {
#pragma unroll
        for (int s = 0; s < NSA; ++s) {
            const int gm = m0 + ((t + 256 * s) / F4R);
            ast[s] = gm < p.M ? p.a_stats[(size_t)gm * p.a_stats_stride] : make_float2(0.f, 0.f);
        }
    }
    const bool k_ragged4 = k_ragged && (p.K & 3) != 0;          // a float4 can straddle K
    const bool n_ragged4 = B_KN && (p.N & 3) != 0;               // ... or N (the [k][n] image)
    // this thread's LDS store slots: one base per operand, the element index an immediate of the ds instruction
    const int sa_off = (t / F4R) * LDS_A + (t % F4R) * 4;                     // + s * (256 / F4R) rows
    const int sb_off = B_KN ? (t >> 5) * LDS_BKN + (t & 31) * 4 : sa_off;     // [k][n] image: + s * 8 k-rows
    auto store_tiles_m = [&](int buf, auto masked_c, int kt, int n0) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_c)::value;
        float* a = As + buf * A_TILE + sa_off;
        float* b = Bs + buf * B_TILE + sb_off;
        const int k0 = kt * BK;
#pragma unroll
        for (int s = 0; s < NSA; ++s) {
            const int c4 = ((t + 256 * s) % F4R) * 4;
            float4 v = ra[s];
            if constexpr (SM) {
                // per element one FMA, v_exp_f32, one multiply; the FMA and the multiply as PACKED fp32 on element pairs (a packed
                // instruction costs the SIMD's fp32 matrix pipe once, tools/ubench/mfma_valu.hip): 8 instead of 12 VALU per float4.
                // Same fp32 operations on the same values: bitwise the scalar form.
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                constexpr float kL2e = 1.4426950408889634f;
                const f32x2 l2 = {kL2e, kL2e}, nm = {-ast[s].x, -ast[s].x}, rs = {ast[s].y, ast[s].y};
                f32x2 a = __builtin_elementwise_fma(f32x2{v.x, v.y}, l2, nm), b = __builtin_elementwise_fma(f32x2{v.z, v.w}, l2, nm);
                a = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} * rs;
                b = f32x2{__builtin_amdgcn_exp2f(b.x), __builtin_amdgcn_exp2f(b.y)} * rs;
                v = make_float4(a.x, a.y, b.x, b.y);
            }
            if constexpr (MASKED) v = mask4(v, k0 + c4, p.K);
            *reinterpret_cast<float4*>(a + s * (256 / F4R) * LDS_A) = v;
        }
        if (!B_KN) {
#pragma unroll
            for (int s = 0; s < NSA; ++s) {
                const int c4 = ((t + 256 * s) % F4R) * 4;
                float4 v = rb[s];
                if constexpr (MASKED) v = mask4(v, k0 + c4, p.K);
                *reinterpret_cast<float4*>(b + s * (256 / F4R) * LDS_A) = v;
            }
        } else {
#pragma unroll
            for (int s = 0; s < NSB; ++s) {
                const int n4 = ((t + 256 * s) & 31) * 4;
                float4 v = rb[s];
                if constexpr (MASKED) v = mask4(v, n0 + n4, p.N);
                *reinterpret_cast<float4*>(b + s * 8 * LDS_BKN) = v;
            }
        }
    };
    auto store_tiles = [&](int buf, int kt, int n0) __attribute__((always_inline)) {
        // ONE uniform branch around the whole store phase: only the last k-tile of a K (column tile of an N) that is not a multiple of 4 masks
        if ((k_ragged4 && kt * BK + BK > p.K) || (n_ragged4 && n0 + BN > p.N)) store_tiles_m(buf, std::true_type{}, kt, n0);
        else store_tiles_m(buf, std::false_type{}, kt, n0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // epilogue: D[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    auto epilogue = [&](int n0) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + li;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row >= p.M) continue;
                    if constexpr (!MULTI) {
                        if (p.splitk > 1) {
                            p.partial[((size_t)blockIdx.y * p.M + row) * p.N + col] = acc[mi][ni][r];
                            continue;
                        }
                    }
                    float v = (acc[mi][ni][r] + bv) * p.alpha;
                    if constexpr (!MULTI) {
                        if (p.act == CIAOSR_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (p.act == CIAOSR_ACT_PRELU) v = v > 0.f ? v : v * p.slope;
                        else if (p.act == CIAOSR_ACT_SIN) v = sinf(v);
                        else if (p.act == CIAOSR_ACT_COS) v = cosf(v);
                    }
                    p.C[(size_t)row * p.ldc + col] = v;
                }
            }
        }
    };

    // flattened (column tile, k-tile) iterations; with tn_per_wg == 1 (and for split-K) this is the plain k-loop
    const int nkl = kt1 - kt0;
    const int total = ntl * nkl;
    load_tiles(kt0, tile0 * BN);
    store_tiles(0, kt0, tile0 * BN);
    __syncthreads();

    int c_tile = tile0, c_kt = kt0;
    for (int i = 0; i < total; ++i) {
        const int cur = i & 1;
        int n_tile = c_tile, n_kt = c_kt + 1;                      // next iteration
        if (n_kt == kt1) { n_kt = kt0; ++n_tile; }
        if (i + 1 < total) load_tiles(n_kt, n_tile * BN);
        const float* a = As + cur * A_TILE + (wm * 64 + li) * LDS_A + 4 * lh;
        const float* b = B_KN ? (Bs + cur * B_TILE + (4 * lh) * LDS_BKN + wn * 64 + li)
                              : (Bs + cur * B_TILE + (wn * 64 + li) * LDS_A + 4 * lh);
        // Fragments of chunk j + 1 are requested BEFORE the 16 MFMAs of chunk j, into the other of two static register stages, and a
        // scheduling barrier keeps them there: left alone hipcc reads a chunk's fragments right in front of its MFMAs (or sinks the
        // requests into the MFMA stream to shorten live ranges), and every 8 MFMAs then wait for an LDS round trip that only the
        // SIMD's other wave can cover (ablation, 192x192 tile: MFMAs + fragment reads alone ran at 0.82 of the matrix peak).
        // Component-major MFMA order: consecutive instructions go to different accumulators.
        float4 fa[2][2], fb[2][2];
        auto read_frags = [&](int j, float4 (&xa)[2], float4 (&xb)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) xa[mi] = *reinterpret_cast<const float4*>(a + mi * 32 * LDS_A + 8 * j);
            if (!B_KN) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) xb[ni] = *reinterpret_cast<const float4*>(b + ni * 32 * LDS_A + 8 * j);
            } else {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const float* bb = b + (8 * j) * LDS_BKN + ni * 32;
                    xb[ni] = make_float4(bb[0], bb[LDS_BKN], bb[2 * LDS_BKN], bb[3 * LDS_BKN]);
                }
            }
        };
        read_frags(0, fa[0], fb[0]);
#pragma unroll
        for (int j = 0; j < BK / 8; ++j) {
            if (j + 1 < BK / 8) read_frags(j + 1, fa[(j + 1) & 1], fb[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#define CIAOSR_GEMM_C(c)                                                                                                  \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                              \
                _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                          \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j & 1][mi].c, fb[j & 1][ni].c, acc[mi][ni], 0, 0, 0);
            CIAOSR_GEMM_C(x) CIAOSR_GEMM_C(y) CIAOSR_GEMM_C(z) CIAOSR_GEMM_C(w)
#undef CIAOSR_GEMM_C
            __builtin_amdgcn_sched_barrier(0);
        }
        if (i + 1 < total) store_tiles(cur ^ 1, n_kt, n_tile * BN);
        if constexpr (MULTI) {
            if (c_kt + 1 == kt1) {                                 // last k-tile of a column tile: its epilogue, then a fresh accumulator
                epilogue(c_tile * BN);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            }
        }
        c_tile = n_tile; c_kt = n_kt;
        __syncthreads();
    }
    if constexpr (!MULTI) epilogue(tile0 * BN);                    // one column tile per workgroup
}

__global__ void gemm_reduce_kernel(GemmP p) {
    const long n = (long)p.M * p.N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += p.partial[(size_t)s * n + i];      // fixed order: deterministic
        const int row = (int)(i / p.N), col = (int)(i - (long)row * p.N);
        v = (v + (p.bias ? p.bias[col] : 0.f)) * p.alpha;
        if (p.act == CIAOSR_ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == CIAOSR_ACT_PRELU) v = v > 0.f ? v : v * p.slope;
        else if (p.act == CIAOSR_ACT_SIN) v = sinf(v);
        else if (p.act == CIAOSR_ACT_COS) v = cosf(v);
        p.C[(size_t)row * p.ldc + col] = v;
    }
}

static int gemm_launch(GemmP& p, bool b_kn, hipStream_t stream, const char* tag);

// skinny GEMM (few output tiles, long K): split the K loop over `partial` ([splits][M][N] floats available)
int gemm_f32_splitk(const float* A, int lda, const float* B, int ldb, bool b_kn, float* C, int ldc, const float* bias,
                    int M, int N, int K, float alpha, int act, float slope, float* partial, size_t partial_floats,
                    hipStream_t stream, const char* tag);

int gemm_f32(const float* A, int lda, const float* B, int ldb, bool b_kn, float* C, int ldc,
             const float* bias, int M, int N, int K, float alpha, int act, float slope,
             hipStream_t stream, const char* tag) {
    if (M <= 0 || N <= 0) return CIAOSR_OK;
    CIAOSR_CHECK_ARG(K > 0 && A && B && C);
    CIAOSR_CHECK_ARG((lda & 3) == 0 && (ldb & 3) == 0);
    CIAOSR_CHECK_ARG(aligned16(A) && aligned16(B));
    if (((size_t)(M - 1) * lda + K) * sizeof(float) >= 0xFFFFFF00ull && M > 1) {
        // A spans more than one buffer descriptor (4 GiB): rows are independent, so run row blocks (multiples of the row tile while
        // there are that many rows; a strided row set -- one row per image line -- may have to go down to single rows)
        const int half = M > BM ? (M / 2 + BM - 1) / BM * BM : (M + 1) / 2;
        const int rc = gemm_f32(A, lda, B, ldb, b_kn, C, ldc, bias, half, N, K, alpha, act, slope, stream, tag);
        if (rc != CIAOSR_OK) return rc;
        return gemm_f32(A + (size_t)half * lda, lda, B, ldb, b_kn, C + (size_t)half * ldc, ldc, bias, M - half, N, K, alpha, act, slope, stream, tag);
    }
    GemmP p;
    p.A = A; p.B = B; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.slope = slope; p.act = act;
    p.a_stats = nullptr; p.a_stats_stride = 0;
    {
        const size_t ab = ((size_t)(M - 1) * lda + K) * sizeof(float);
        const size_t bb = (b_kn ? ((size_t)(K - 1) * ldb + N) : ((size_t)(N - 1) * ldb + K)) * sizeof(float);
        CIAOSR_CHECK_ARG(ab < 0xFFFFFF00ull && bb < 0xFFFFFF00ull);   // 32-bit buffer offsets
        p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    }
    p.tiles_n = ceil_div(N, BN);
    p.splitk = 1;
    p.kt_per_split = (K + BK - 1) / BK;
    p.partial = nullptr;
    // short K (the correlation scores: K = 288 = 9 k-tiles against 72 column tiles): a few consecutive column tiles per workgroup as one
    // software pipeline, as long as the grid still fills the chip several times over.  192x192 tile: 2.50 -> 1.84 ms with 3 tiles per
    // workgroup (0.50 -> 0.68 of the fp32 MFMA peak; 6 tiles 1.88, 12 tiles 2.03: the epilogue of a tile is serial work of its wave)
    int tn = 1;
#ifndef CIAOSR_GEMM32_TN1
    if (p.kt_per_split <= 24 && act == CIAOSR_ACT_NONE) {
        tn = ceil_div(CIAOSR_GEMM32_TNK, p.kt_per_split);
        while (tn > 1 && (long)ceil_div(M, BM) * ceil_div(p.tiles_n, tn) < 2048) --tn;
    }
#endif
    p.tn_per_wg = tn; p.groups_n = ceil_div(p.tiles_n, tn);
    p.n_wg = ceil_div(M, BM) * p.groups_n;
    return gemm_launch(p, b_kn, stream, tag);
}

static int gemm_launch(GemmP& p, bool b_kn, hipStream_t stream, const char* tag) {
    const size_t smem = (size_t)(2 * A_TILE + 2 * B_TILE) * sizeof(float);
    CIAOSR_BIG_LDS((gemm_f32_kernel<true, false>), smem);
    CIAOSR_BIG_LDS((gemm_f32_kernel<false, false>), smem);
    CIAOSR_BIG_LDS((gemm_f32_kernel<true, true>), smem);
    CIAOSR_BIG_LDS((gemm_f32_kernel<false, true>), smem);
    CIAOSR_BIG_LDS((gemm_f32_kernel<true, false, true>), smem);
    CIAOSR_BIG_LDS((gemm_f32_kernel<false, false, true>), smem);
    {
        ProfScope prof(tag ? tag : (b_kn ? "gemm_f32_nn" : "gemm_f32_nt"), stream);
        const dim3 grid(p.n_wg, p.splitk);
        if (p.a_stats) {                                         // gemm_f32_softmax_a: one column tile per workgroup
            if (b_kn) hipLaunchKernelGGL((gemm_f32_kernel<true, false, true>), grid, dim3(256), smem, stream, p);
            else hipLaunchKernelGGL((gemm_f32_kernel<false, false, true>), grid, dim3(256), smem, stream, p);
        } else if (p.tn_per_wg > 1) {
            if (b_kn) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), smem, stream, p);
            else hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), smem, stream, p);
        } else {
            if (b_kn) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), smem, stream, p);
            else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), smem, stream, p);
        }
    }
    int rc = launch_status("gemm_f32");
    if (rc != CIAOSR_OK || p.splitk == 1) return rc;
    ProfScope prof("gemm_splitk_reduce", stream);
    const long n = (long)p.M * p.N;
    int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3(grid > 2048 ? 2048 : grid), dim3(256), 0, stream, p);
    return launch_status("gemm_reduce");
}

int gemm_f32_splitk(const float* A, int lda, const float* B, int ldb, bool b_kn, float* C, int ldc, const float* bias,
                    int M, int N, int K, float alpha, int act, float slope, float* partial, size_t partial_floats,
                    hipStream_t stream, const char* tag) {
    if (M <= 0 || N <= 0) return CIAOSR_OK;
    CIAOSR_CHECK_ARG(K > 0 && A && B && C && partial);
    CIAOSR_CHECK_ARG((lda & 3) == 0 && (ldb & 3) == 0 && aligned16(A) && aligned16(B));
    GemmP p;
    p.A = A; p.B = B; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.slope = slope; p.act = act;
    p.a_stats = nullptr; p.a_stats_stride = 0;
    const size_t ab = ((size_t)(M - 1) * lda + K) * sizeof(float);
    const size_t bb = (b_kn ? ((size_t)(K - 1) * ldb + N) : ((size_t)(N - 1) * ldb + K)) * sizeof(float);
    CIAOSR_CHECK_ARG(ab < 0xFFFFFF00ull && bb < 0xFFFFFF00ull);
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    p.tiles_n = ceil_div(N, BN);
    p.tn_per_wg = 1; p.groups_n = p.tiles_n;
    p.n_wg = ceil_div(M, BM) * p.tiles_n;
    const int nk = (K + BK - 1) / BK;
    int splits = ceil_div(512, p.n_wg);
    if (splits > nk / 4) splits = nk / 4 > 0 ? nk / 4 : 1;
    if ((size_t)splits * M * N > partial_floats) splits = (int)(partial_floats / ((size_t)M * N));
    if (splits < 1) splits = 1;
    p.kt_per_split = ceil_div(nk, splits);
    p.splitk = ceil_div(nk, p.kt_per_split);
    p.partial = partial;
    return gemm_launch(p, b_kn, stream, tag);
}

int gemm_f32_softmax_a(const float* A, int lda, const float* a_stats2, int a_stats_stride, const float* B, int ldb, bool b_kn, float* C, int ldc,
                       int M, int N, int K, float* partial, size_t partial_floats, hipStream_t stream, const char* tag) {
    if (M <= 0 || N <= 0) return CIAOSR_OK;
    CIAOSR_CHECK_ARG(K > 0 && A && B && C && a_stats2 && a_stats_stride >= 1);
    CIAOSR_CHECK_ARG((lda & 3) == 0 && (ldb & 3) == 0 && aligned16(A) && aligned16(B) && (reinterpret_cast<uintptr_t>(a_stats2) & 7u) == 0);
    if (((size_t)(M - 1) * lda + K) * sizeof(float) >= 0xFFFFFF00ull && M > 1) {       // A past one descriptor: row blocks (see gemm_f32)
        const int half = M > BM ? (M / 2 + BM - 1) / BM * BM : (M + 1) / 2;
        const int rc = gemm_f32_softmax_a(A, lda, a_stats2, a_stats_stride, B, ldb, b_kn, C, ldc, half, N, K, partial, partial_floats, stream, tag);
        if (rc != CIAOSR_OK) return rc;
        return gemm_f32_softmax_a(A + (size_t)half * lda, lda, a_stats2 + (size_t)half * a_stats_stride * 2, a_stats_stride, B, ldb, b_kn,
                                  C + (size_t)half * ldc, ldc, M - half, N, K, partial, partial_floats, stream, tag);
    }
    GemmP p;
    p.A = A; p.B = B; p.C = C; p.bias = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = 1.f; p.slope = 0.f; p.act = CIAOSR_ACT_NONE;
    p.a_stats = reinterpret_cast<const float2*>(a_stats2); p.a_stats_stride = a_stats_stride;
    const size_t ab = ((size_t)(M - 1) * lda + K) * sizeof(float);
    const size_t bb = (b_kn ? ((size_t)(K - 1) * ldb + N) : ((size_t)(N - 1) * ldb + K)) * sizeof(float);
    CIAOSR_CHECK_ARG(ab < 0xFFFFFF00ull && bb < 0xFFFFFF00ull);
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    p.tiles_n = ceil_div(N, BN);
    p.tn_per_wg = 1; p.groups_n = p.tiles_n;
    p.n_wg = ceil_div(M, BM) * p.tiles_n;
    const int nk = (K + BK - 1) / BK;
    p.splitk = 1; p.kt_per_split = nk; p.partial = nullptr;
    if (partial) {                                                   // skinny problem: the split of gemm_f32_splitk
        int splits = ceil_div(512, p.n_wg);
        if (splits > nk / 4) splits = nk / 4 > 0 ? nk / 4 : 1;
        if ((size_t)splits * M * N > partial_floats) splits = (int)(partial_floats / ((size_t)M * N));
        if (splits < 1) splits = 1;
        p.kt_per_split = ceil_div(nk, splits);
        p.splitk = ceil_div(nk, p.kt_per_split);
        p.partial = partial;
    }
    return gemm_launch(p, b_kn, stream, tag);
}

}  // namespace ciaosr

extern "C" int ciaosr_gemm_f32(const float* A, int lda, const float* B, int ldb, int b_is_kn, float* C,
                               int ldc, const float* bias, int M, int N, int K, float alpha, int act,
                               float slope, void* stream) {
    return ciaosr::gemm_f32(A, lda, B, ldb, b_is_kn != 0, C, ldc, bias, M, N, K, alpha, act, slope,
                            (hipStream_t)stream, nullptr);
}
