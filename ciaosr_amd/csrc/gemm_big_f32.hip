// attn.V of CrossScaleAttention at the size of a C3 tile (arch_csnln.py:511-516: softmax over the key axis, then the transposed
// convolution with the value patches = P[M x K] . V'[K x N]) with ONE 256-thread workgroup per CU and a 192 x 256 x 16 tile (round 4).
//
// gemm_f32.hip's 128 x 128 tile with three workgroups per CU measures 0.785 of the fp32 MFMA peak at this shape (36 864 x 1024 x 9216)
// with or without the softmax in its staging; its ablation (MFMAs + fragment reads alone: 0.92) and the vendor library's plain GEMM at
// the same shape (0.936) say that the loss is the tile, not the arithmetic.  What dense_wino4_f32.hip taught about one wave per SIMD is
// applied here:
//   * wave tile 96 x 128 = 3 x 4 MFMA tiles (192 accumulator registers): 3 + 4 fragments feed 12 MFMAs per k-step instead of 2 + 2 feeding
//     4; 768 workgroup tiles at the C3 size = exactly 3 rounds of the 256 CUs;
//   * every memory instruction of a k-tile sits behind ONE MFMA (the in-order wave issues it inside that MFMA's 64 pipe cycles): the
//     fragments of the next 8-deep chunk, the global loads of the k-tile after the next;
//   * the VALU work (exp2 of the logits, the staging's LDS writes) is separated in TIME from the MFMAs -- a VALU instruction beside an fp32
//     MFMA costs a lone wave ~20 pipe cycles (tools/ubench/mfma_valu.hip) --: a short T phase per k-tile, then one barrier;
//   * THREE LDS buffers: the k-tile written in a T phase is read two M phases later, so the first fragments of the next k-tile are read
//     during the current one (no LDS round trip behind the barrier).
// Same products summed in the same order as gemm_f32_kernel<true, false, true> (k = 8 j + 4 h + e per MFMA e of chunk j; one accumulator
// per output, k ascending): bitwise the same result, which is what the test asserts.
#include "ops.h"
#include <type_traits>

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int GBM = 192, GBN = 256, GBK = 16;
constexpr int GB_LDA = GBK + 4;                 // 20 floats per A row (conflict-free b128 fragment reads)
constexpr int GB_LDB = GBN + 64;                // 320 floats per k-row of the [k][n] image of B: 5 x 256 B, so that a B fragment's four k-rows
                                                // are two ds_read2st64_b32
constexpr int GB_AT = GBM * GB_LDA;             // 3840 floats
constexpr int GB_BT = GBK * GB_LDB;             // 5120 floats
constexpr int GB_BUF = GB_AT + GB_BT;           // one buffer: 8960 floats
constexpr size_t kGemmBigLds = 3 * (size_t)GB_BUF * sizeof(float);      // 107 520 B
constexpr unsigned kOobGB = 0xFFFFFFF0u;

struct GemmBigP {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    unsigned a_bytes, b_bytes;
    const float2* a_stats; int a_stats_stride;   // row statistics of the logits: (max x log2 e, 1 / sum of exp)
    int tiles_n, n_wg;
};

__device__ __forceinline__ float4 gb_ld4s(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, unsigned s_off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, (int)s_off, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

__global__ __launch_bounds__(256) void gemm_big_softmax_f32_kernel(GemmBigP p) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    // XCD-aware bijective remap (as gemm_f32_kernel): an XCD walks a contiguous run of tiles, consecutive tiles share the A row block
    const int bid = blockIdx.x;
    const int q8 = p.n_wg >> 3, r8 = p.n_wg & 7;
    const int xcd = bid & 7, slot0 = bid >> 3;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot0;
    const int m0 = (lid / p.tiles_n) * GBM, n0 = (lid % p.tiles_n) * GBN;

    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);

    // this thread's 3 float4 of an A tile (row = idx / 4, k = 4 (idx % 4)) and 4 float4 of a B tile (k = idx / 64, n = 4 (idx % 64))
    unsigned offA[3], offB[4];
    float2 ast[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int idx = t + 256 * s, gm = m0 + (idx >> 2);
        offA[s] = gm < p.M ? ((unsigned)gm * (unsigned)p.lda + (unsigned)((idx & 3) * 4)) * 4u : kOobGB;
        ast[s] = gm < p.M ? p.a_stats[(size_t)gm * p.a_stats_stride] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = t + 256 * s, gn = n0 + (idx & 63) * 4;
        offB[s] = gn < p.N ? ((unsigned)(idx >> 6) * (unsigned)p.ldb + (unsigned)gn) * 4u : kOobGB;
    }
    const int sa_off = (t >> 2) * GB_LDA + (t & 3) * 4;            // + 64 s rows
    const int sb_off = (t >> 6) * GB_LDB + (t & 63) * 4;           // + 4 s k-rows
    const int nk = p.K / GBK;                                       // K is a multiple of 48 (launcher)

    float4 ra[3], rb[4];
    auto load_tiles = [&](int kt) __attribute__((always_inline)) {
        const unsigned ska = (unsigned)(kt * GBK) * 4u, skb = (unsigned)(kt * GBK) * (unsigned)p.ldb * 4u;
#pragma unroll
        for (int s = 0; s < 3; ++s) ra[s] = gb_ld4s(rs_a, offA[s], ska);
#pragma unroll
        for (int s = 0; s < 4; ++s) rb[s] = gb_ld4s(rs_b, offB[s], skb);
    };
    // T phase: probabilities out of the logits (one FMA, v_exp_f32, one multiply per element; FMA and multiply packed on element pairs:
    // the same fp32 operations as gemm_f32_kernel's staging), then their LDS writes.  B's four float4 need no arithmetic: they are
    // written from the M phase's slots (store_b below).
    auto store_a = [&](float* a) __attribute__((always_inline)) {
        constexpr float kL2e = 1.4426950408889634f;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const float4 v = ra[s];
            const f32x2 l2 = {kL2e, kL2e}, nm = {-ast[s].x, -ast[s].x}, rs = {ast[s].y, ast[s].y};
            f32x2 x = __builtin_elementwise_fma(f32x2{v.x, v.y}, l2, nm), y = __builtin_elementwise_fma(f32x2{v.z, v.w}, l2, nm);
            x = f32x2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)} * rs;
            y = f32x2{__builtin_amdgcn_exp2f(y.x), __builtin_amdgcn_exp2f(y.y)} * rs;
            *reinterpret_cast<float4*>(a + s * 64 * GB_LDA) = make_float4(x.x, x.y, y.x, y.y);
        }
    };
    auto store_b = [&](float* b, int s) __attribute__((always_inline)) { *reinterpret_cast<float4*>(b + s * 4 * GB_LDB) = rb[s]; };

    f32x16 acc[3][4];
#pragma unroll
    for (int mi = 0; mi < 3; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // Per-lane base pointers of the three buffers (fragment reads and staging writes): the k loop is unrolled three times, so that the
    // buffer a k-tile uses is known at compile time -- reads and writes are base register + immediate, and nothing rotates.
    const float* fa_b[3]; const float* fb_b[3]; float* sa_b[3]; float* sb_b[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        fa_b[u] = gsm + u * GB_BUF + (wm * 96 + li) * GB_LDA + 4 * lh;
        fb_b[u] = gsm + u * GB_BUF + GB_AT + (4 * lh) * GB_LDB + wn * 128 + li;
        sa_b[u] = gsm + u * GB_BUF + sa_off;
        sb_b[u] = gsm + u * GB_BUF + GB_AT + sb_off;
    }
    // fragment reads: chunk j (8 k's) of a buffer: lane (i, h) holds k = 8 j + 4 h + e for MFMA e
    float4 fa[2][3], fb[2][4];
    auto read_a = [&](const float* base, int j, int mi, float4& x) __attribute__((always_inline)) {
        x = *reinterpret_cast<const float4*>(base + mi * 32 * GB_LDA + 8 * j);
    };
    auto read_b = [&](const float* base, int j, int ni, float4& x) __attribute__((always_inline)) {
        const float* bb = base + (8 * j) * GB_LDB + ni * 32;
        x = make_float4(bb[0], bb[GB_LDB], bb[2 * GB_LDB], bb[3 * GB_LDB]);
    };

    // prologue: k-tiles 0 and 1 into buffers 0 and 1; chunk 0 of k-tile 0 in registers
    load_tiles(0);
    store_a(sa_b[0]);
#pragma unroll
    for (int s = 0; s < 4; ++s) store_b(sb_b[0], s);
    load_tiles(1);
    store_a(sa_b[1]);
#pragma unroll
    for (int s = 0; s < 4; ++s) store_b(sb_b[1], s);
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 3; ++mi) read_a(fa_b[0], 0, mi, fa[0][mi]);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) read_b(fb_b[0], 0, ni, fb[0][ni]);

    // One iteration = k-tile kt out of buffer CUR (= kt % 3), k-tile kt + 1 in buffer NXT (written in T(kt - 1)), k-tile kt + 2 goes
    // to buffer WR (it held k-tile kt - 1).
    //   M phase: 2 chunks x 4 components x 12 tiles = 96 MFMAs, a slot's memory instruction right behind its MFMA:
    //     chunk 0: the fragments of chunk 1 of this k-tile; the 7 global loads of k-tile kt + 2 (half an M phase to land);
    //     the ONE barrier of the iteration;
    //     chunk 1: the fragments of chunk 0 of k-tile kt + 1; B's four staging writes of k-tile kt + 2.
    //   T phase: exp2 of A's three float4 and their writes.  It runs straight into the next M phase: the next chunk-0 fragments are already
    //   in registers.
    // Where the barrier stands: k-tile kt + 1 was written in chunk 1 of M(kt - 1) (B) and T(kt - 1) (A); its first reads are in chunk 1's
    // slots, behind the barrier every wave reaches only after its T(kt - 1).  Buffer WR's last reads (k-tile kt - 1) were in chunk 0's
    // slots of M(kt - 1): every wave that is past THIS iteration's barrier has left them behind.
    auto iteration = [&](int kt, auto cur_c) __attribute__((always_inline)) {
        constexpr int CUR = decltype(cur_c)::value, NXT = (CUR + 1) % 3, WR = (CUR + 2) % 3;
        const unsigned kk = (unsigned)((kt + 2 < nk ? kt + 2 : nk - 1) * GBK);       // past the end: the last k-tile again, not stored
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1) {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int sl = 0; sl < 48; ++sl) {
                const int c = sl / 12, mi = (sl % 12) / 4, ni = sl % 4;
                const float4 xa = fa[j][mi], xb = fb[j][ni];
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(c == 0 ? xa.x : c == 1 ? xa.y : c == 2 ? xa.z : xa.w,
                                                                    c == 0 ? xb.x : c == 1 ? xb.y : c == 2 ? xb.z : xb.w, acc[mi][ni], 0, 0, 0);
                if (sl % 6 == 1 && sl / 6 < 7) {                    // the next chunk's fragments: 7 slots of 48
                    const int q = sl / 6;
                    if (q < 3) read_a(j == 0 ? fa_b[CUR] : fa_b[NXT], j == 0 ? 1 : 0, q, fa[j ^ 1][q]);
                    else read_b(j == 0 ? fb_b[CUR] : fb_b[NXT], j == 0 ? 1 : 0, q - 3, fb[j ^ 1][q - 3]);
                }
                if (j == 0 && sl % 6 == 4 && sl / 6 < 7) {          // the global loads of k-tile kt + 2
                    const int q = sl / 6;
                    if (q < 3) ra[q] = gb_ld4s(rs_a, offA[q], kk * 4u);
                    else rb[q - 3] = gb_ld4s(rs_b, offB[q - 3], kk * (unsigned)p.ldb * 4u);
                }
                if (j == 1 && sl % 6 == 4 && sl / 6 >= 3 && sl / 6 < 7) store_b(sb_b[WR], sl / 6 - 3);   // harmless past the end: nobody reads it
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        store_a(sa_b[WR]);
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int kt = 0; kt < nk; kt += 3) {                            // nk is a multiple of 3 (launcher)
        iteration(kt, std::integral_constant<int, 0>{});
        iteration(kt + 1, std::integral_constant<int, 1>{});
        iteration(kt + 2, std::integral_constant<int, 2>{});
    }

    // epilogue: D[row][col], col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int col = n0 + wn * 128 + ni * 32 + li;
        if (col >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < 3; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 96 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < p.M) p.C[(size_t)row * p.ldc + col] = acc[mi][ni][r];
            }
    }
}

// true when the big-tile kernel takes the problem (else the caller keeps gemm_f32_softmax_a's 128 x 128 kernel)
bool gemm_big_softmax_f32_ok(int lda, int ldb, int M, int N, int K, bool b_kn) {
    return b_kn && (K % (3 * GBK)) == 0 && (N & 3) == 0 && (lda & 3) == 0 && (ldb & 3) == 0 &&
           (long)ceil_div(M, GBM) * ceil_div(N, GBN) >= 256;       // at least one workgroup per CU
}

int gemm_big_softmax_f32(const float* A, int lda, const float* a_stats2, int a_stats_stride, const float* B, int ldb, float* C, int ldc,
                         int M, int N, int K, hipStream_t stream, const char* tag) {
    CIAOSR_CHECK_ARG(gemm_big_softmax_f32_ok(lda, ldb, M, N, K, true) && A && B && C && a_stats2 && aligned16(A) && aligned16(B));
    GemmBigP p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    const size_t ab = ((size_t)(M - 1) * lda + K) * sizeof(float), bb = ((size_t)(K - 1) * ldb + N) * sizeof(float);
    CIAOSR_CHECK_ARG(ab < 0xFFFFFF00ull && bb < 0xFFFFFF00ull);
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    p.a_stats = reinterpret_cast<const float2*>(a_stats2); p.a_stats_stride = a_stats_stride;
    p.tiles_n = ceil_div(N, GBN);
    p.n_wg = ceil_div(M, GBM) * p.tiles_n;
    CIAOSR_BIG_LDS(gemm_big_softmax_f32_kernel, kGemmBigLds);
    ProfScope prof(tag ? tag : "gemm_big_softmax", stream);
    hipLaunchKernelGGL(gemm_big_softmax_f32_kernel, dim3(p.n_wg), dim3(256), kGemmBigLds, stream, p);
    return launch_status("gemm_big_softmax_f32");
}

}  // namespace ciaosr
