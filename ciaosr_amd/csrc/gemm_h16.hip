// 16-bit-MFMA NT GEMM (precision modes "bf16" / "f16" of CrossScaleAttention's two big contractions; compiled once per
// element type, h16_util.h -- "bf16" below = the 16-bit element type of the build):
//   C[M][N] = alpha * A[M][K] . B[N][K]^T,  A and B 16-bit in memory, fp32 accumulation, C fp32 or 16-bit.
// Used for the correlation scores Q.K^T (arch_csnln.py:497-499, K = 9C/2) and the attention-weighted patch sum
// P.V' (arch_csnln.py:511 in the composed form of patch_ops.hip, K = L) when the host asks for bf16.
//
// v_mfma_f32_32x32x16_bf16 is 16x the fp32 MFMA rate, so the tile is sized for the operand streams instead:
// workgroup tile 256 x 128 x 32, 4 waves in a 2 x 2 grid, each wave a 128 x 64 sub-tile = 4 x 2 MFMA tiles (128
// accumulator registers); the k-tile is 32 deep so that TWO workgroups share a CU (with one wave per SIMD -- round 1: 108 KB,
// one workgroup per CU -- every barrier and late load stalled the matrix pipe directly: MFMA pipe 27 % busy).
// LDS rows are 64 B, unpadded, with the 16-byte chunk index XOR-swizzled by (row >> 2) & 3: the 16 rows a ds_read_b128 lane
// group touches ({0-3,12-15,20-27} / {4-11,16-19,28-31}) land on 16 distinct bank quads.
// Staging (round 2, second half): LDS-DMA.  The register-staged version (global -> VGPR two k-tiles ahead -> ds_write_b128) paid
// 311 LDS cycles per 24-KB k-tile for the VGPR -> LDS transfer (~79 B/clk per CU, MI355X_MICROARCH.md LDS table) on top of 192 for
// the fragment reads, against 512 MFMA cycles per SIMD: with two workgroups per CU the LDS pipe was as busy as the matrix pipe.
// Because the LDS image is lane-linear per wave (unpadded rows, swizzle), each wave's 1-KB slice of a stage now comes straight from
// memory (`buffer_load_dwordx4 ... lds`), the swizzle applied to the SOURCE address; three stage buffers (72 KB), two k-tiles in
// flight across every barrier (counted s_waitcnt vmcnt + raw s_barrier), 48 VGPRs fewer.  P.V' of the 192x192 tile: 0.90 -> 0.79 ms.
// Swapped MFMA operands (B rows = A operand, A rows = B operand): a lane owns one output row and 4 consecutive
// columns per accumulator quad, so the epilogue stores 16 B (fp32) / 8 B (bf16) pieces.
// Out-of-range rows / k-chunks are buffer loads with an out-of-range offset (return 0, no branches).
#include "h16_util.h"
#include "ops.h"

namespace ciaosr {

bool softmax_rows_reg_h16(const float* S, long rows, int L, int ld, unsigned short* P, int ldp, bool f16, hipStream_t s);   // patch_ops.hip

namespace CIAOSR_H16_NS {

constexpr bool kF16 = CIAOSR_F16 != 0;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int GM = 256, GN = 128, GK = 32;               // GM: the default row tile (MT = 4 MFMA tiles per wave); MT = 3 -> 192 rows
constexpr int GRS = GK * 2;                               // LDS row stride in bytes (32 bf16, chunk-swizzled, no pad)
constexpr int GCH = GK / 8;                               // 16-byte chunks per row
constexpr int GSB = GN * GCH / 256;                       // B staging chunks per thread (2); A: MT
constexpr int GB_T = GN * GRS;                            // bytes of B per stage
constexpr size_t kGemm16Lds = 3 * (size_t)(GM * GRS + GB_T);  // three stages of the 256-row tile, 73 728 B: two workgroups per CU
constexpr int GPR = GN * 2 + 16;                          // row pitch of the 16-bit output tile staged in LDS by the softmax epilogue
static_assert((size_t)GM * GPR <= kGemm16Lds, "the staged output tile must fit the stage buffers");
constexpr unsigned kOob16 = 0xFFFFFFF0u;

struct Gemm16P {
    const unsigned short* A; int lda;
    const unsigned short* B; int ldb;
    unsigned a_bytes, b_bytes;
    void* C; int ldc; int c_bf16;
    int M, N, K;
    float alpha;
    int tiles_n, n_wg;
    // optional epilogue (full-width quads only use these; all null for the plain GEMM): v = acc * alpha + bias[n] + res[m][n],
    // written to C and, when given, to a second fp32 matrix C2 and a 16-bit matrix C16
    const float* bias;
    const float* res; int ldres;
    float* C2; int ldc2;
    unsigned short* C16; int ldc16;
    // row-softmax epilogues (EPI 1 / 2): the logits alpha * A.B^T never go to memory.
    //   EPI 1: per (row, 64-column wave strip) partial (max, sum of exp(v - max)) -> part[row][2 * tile_n + wn]
    //   EPI 2: stats[row] = (row max, 1 / row sum) -> probabilities exp(v - max) / sum as 16-bit into C16 (columns [N, npad) zeroed)
    float2* part; int n_part;
    const float2* stats; int npad;
    int relu = 0;                    // fused epilogue (bias given): v = max(v, 0) before it is written (staged MLP layers)
};

// store epilogue of one 256 x 128 tile: accumulator quad q of (mt, nt) = row m0 + 128 wm + 32 mt + li, columns n0 + 64 wn + 32 nt + 8 q + 4 lh .. +3
template <int MT>
__device__ __forceinline__ void store_tile(const Gemm16P& p, const f32x16 (&acc)[MT][2], int m0, int n0, int wm, int wn, int li, int lh) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + 32 * MT * wm + 32 * mt + li;
        if (m >= p.M) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + 64 * wn + 32 * nt + 8 * q + 4 * lh;
                if (n >= p.N) continue;
                float v0 = acc[mt][nt][4 * q] * p.alpha, v1 = acc[mt][nt][4 * q + 1] * p.alpha;
                float v2 = acc[mt][nt][4 * q + 2] * p.alpha, v3 = acc[mt][nt][4 * q + 3] * p.alpha;
                if (p.bias) {                                              // fused epilogue (host checks N % 4 == 0: whole quads)
                    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                    v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
                    if (p.res) {
                        const float4 r = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldres + n);
                        v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
                    }
                    if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                    if (p.C2) *reinterpret_cast<float4*>(p.C2 + (size_t)m * p.ldc2 + n) = make_float4(v0, v1, v2, v3);
                    if (p.C16) *reinterpret_cast<uint2*>(p.C16 + (size_t)m * p.ldc16 + n) = pack_h16x4<kF16>(v0, v1, v2, v3);
                }
                if (n + 3 >= p.N) {                                        // ragged last columns
                    const float v[4] = {v0, v1, v2, v3};
                    for (int e = 0; e < 4 && n + e < p.N; ++e) {
                        if (p.c_bf16) reinterpret_cast<unsigned short*>(p.C)[(size_t)m * p.ldc + n + e] = to_h16<kF16>(v[e]);
                        else reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n + e] = v[e];
                    }
                } else if (p.c_bf16)
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.C) + (size_t)m * p.ldc + n) =
                        pack_h16x4<kF16>(v0, v1, v2, v3);
                else
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = make_float4(v0, v1, v2, v3);
            }
    }
}

// softmax pass 1: partial (max, sum of exp) of this wave's 64-column strip.  alpha > 0 (checked by the launcher): the maximum is
// taken on the raw accumulators and exp(alpha x - m) = exp2(x (alpha log2 e) - m log2 e) is one FMA + v_exp_f32 per value -- this
// epilogue is VALU-bound (128 values per lane against 144 MFMAs per wave at K = 288), 10 -> 3 VALU operations per value; interior
// tiles skip the column predicates altogether.
template <int MT>
__device__ __forceinline__ void stats_tile(const Gemm16P& p, const f32x16 (&acc)[MT][2], int m0, int n0, int wm, int wn, int li, int lh) {
    constexpr float kLog2e = 1.4426950408889634f;
    const float c1 = p.alpha * kLog2e;
    const bool full = n0 + GN <= p.N;                      // uniform
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + 32 * MT * wm + 32 * mt + li;
        float mx = -INFINITY, sum = 0.f;
        if (full) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, acc[mt][nt][e]);
            const float ms = mx * c1;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += __builtin_amdgcn_exp2f(__builtin_fmaf(acc[mt][nt][e], c1, -ms));
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n0 + 64 * wn + 32 * nt + 8 * q + 4 * lh + e < p.N) mx = fmaxf(mx, acc[mt][nt][4 * q + e]);
            const float ms = mx * c1;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n0 + 64 * wn + 32 * nt + 8 * q + 4 * lh + e < p.N)
                            sum += __builtin_amdgcn_exp2f(__builtin_fmaf(acc[mt][nt][4 * q + e], c1, -ms));
        }
        mx *= p.alpha;                                     // the statistics are kept in logit units
        const float mo = __shfl_xor(mx, 32, 64), so = __shfl_xor(sum, 32, 64);
        const float mm = fmaxf(mx, mo);
        // a strip with no valid column on either lane keeps (-inf, 0): exp(-inf - -inf) is avoided
        const float tot = (mx == -INFINITY ? 0.f : sum * __expf(mx - mm)) + (mo == -INFINITY ? 0.f : so * __expf(mo - mm));
        if (lh == 0 && m < p.M) p.part[(size_t)m * p.n_part + 2 * (n0 / GN) + wn] = make_float2(mm, tot);
    }
}

// softmax pass 2: probabilities exp(v - row max) / row sum as 16-bit, pad columns zeroed.  The tile goes through LDS (the stage
// buffers are free after the k-loop) so that it leaves in full 256-byte row pieces: written straight from the accumulator layout a
// lane stores 8 bytes per row for 32 different rows per instruction -- 16-byte fragments of 18-KB-apart rows, 680 MB per 192x192
// tile at 1.6 TB/s, the whole pass at 0.18 of the MFMA peak.
template <int MT>
__device__ __forceinline__ void softmax_tile(const Gemm16P& p, const f32x16 (&acc)[MT][2], int m0, int n0, int wm, int wn, int li, int lh,
                                             unsigned char* lds, int t) {
    constexpr float kLog2e = 1.4426950408889634f;
    const float c1 = p.alpha * kLog2e;
    __syncthreads();                                       // every wave is done with the last stage
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = 32 * MT * wm + 32 * mt + li, m = m0 + r;
        const float2 st = m < p.M ? p.stats[m] : make_float2(0.f, 0.f);
        const float ms = st.x * kLog2e;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 64 * wn + 32 * nt + 8 * q + 4 * lh, n = n0 + c;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = n + e < p.N ? __builtin_amdgcn_exp2f(__builtin_fmaf(acc[mt][nt][4 * q + e], c1, -ms)) * st.y : 0.f;
                *reinterpret_cast<uint2*>(lds + r * GPR + c * 2) = pack_h16x4<kF16>(v[0], v[1], v[2], v[3]);
            }
    }
    __syncthreads();
    // 16 chunks of 16 bytes per row: a wave instruction stores 4 whole rows of the tile
#pragma unroll
    for (int u = 0; u < MT * 4; ++u) {
        const int idx = t + 256 * u, r = idx >> 4, ch = idx & 15;
        const int m = m0 + r, n = n0 + 8 * ch;
        if (m < p.M && n < p.npad)
            *reinterpret_cast<uint4*>(p.C16 + (size_t)m * p.ldc16 + n) = *reinterpret_cast<const uint4*>(lds + r * GPR + ch * 16);
    }
}

// ---- the kernel: one 256 x 128 tile per workgroup, operands staged by LDS-DMA, epilogue by EPI -------------------------------
// lane i of a wave's 1-KB slice writes LDS chunk i and fetches the logical chunk (i % 4) ^ key(row); hipcc's __syncthreads() would
// drain the DMAs, hence the raw barrier and the counted waits.
template <int EPI, int MT>
__global__ __launch_bounds__(256, 2) void gemm_h16_kernel(Gemm16P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];      // [3] x { A [GM_][GRS], B [GN][GRS] }
    constexpr int GM_ = 64 * MT;                           // rows of this tile: 2 wave rows x MT MFMA tiles
    constexpr int GSA = MT, GA_T = GM_ * GRS;
    constexpr int STG = GA_T + GB_T;
    const int bid = blockIdx.x;
    const int q8 = p.n_wg >> 3, r8 = p.n_wg & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int m0 = (lid / p.tiles_n) * GM_, n0 = (lid % p.tiles_n) * GN;

    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;

    // DMA slices: wave w, step s covers rows 16 w + 64 s .. + 15 (1 KB of LDS); lane -> row +lane / 4, LDS chunk lane % 4,
    // logical (source) chunk (lane % 4) ^ ((row >> 2) & 3)
    const int rl = lane >> 2, pc = lane & 3;
    unsigned a_off[GSA], b_off[GSB];
#pragma unroll
    for (int s = 0; s < GSA; ++s) {
        const int row = 16 * w + 64 * s + rl, gm = m0 + row;
        const int part = pc ^ ((row >> 2) & 3);
        a_off[s] = gm < p.M ? (unsigned)gm * (unsigned)p.lda * 2u + (unsigned)part * 16u : kOob16;
    }
#pragma unroll
    for (int s = 0; s < GSB; ++s) {
        const int row = 16 * w + 64 * s + rl, gn = n0 + row;
        const int part = pc ^ ((row >> 2) & 3);
        b_off[s] = gn < p.N ? (unsigned)gn * (unsigned)p.ldb * 2u + (unsigned)part * 16u : kOob16;
    }
    const int nk = (p.K + GK - 1) / GK;
    // The DMAs are issued from inline asm: a `__builtin_amdgcn_raw_ptr_buffer_load_lds` is counted by hipcc, which then waits
    // vmcnt(0) in front of the first ds_read of every k-tile (it cannot tell the stage being read from the stages in flight) and the
    // pipeline collapses to depth 0.  Hidden from the compiler, the DMAs are waited for by the counted s_waitcnt below.  M0 (the
    // LDS destination base of the wave's 1-KB slice) is written in the same statement that uses it and restored after.
    const i32x4 da = {(int)(unsigned)(size_t)p.A, (int)(((size_t)p.A >> 32) & 0xFFFFu), (int)p.a_bytes, 0x00020000};
    const i32x4 db = {(int)(unsigned)(size_t)p.B, (int)(((size_t)p.B >> 32) & 0xFFFFu), (int)p.b_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds16;
    auto dma = [&](const i32x4& desc, unsigned lds_dst, unsigned voff) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(desc) : "memory");
    };
    auto issue = [&](int kt, int buf) {
        const int k0 = kt * GK;
        const unsigned st = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * STG) + (unsigned)(16 * w) * GRS);
#pragma unroll
        for (int s = 0; s < GSA; ++s) {
            const int part = pc ^ (((16 * w + 64 * s + rl) >> 2) & 3);
            const bool ok = a_off[s] != kOob16 && k0 + part * 8 < p.K;
            dma(da, st + (unsigned)(64 * s * GRS), ok ? a_off[s] + (unsigned)k0 * 2u : kOob16);
        }
#pragma unroll
        for (int s = 0; s < GSB; ++s) {
            const int part = pc ^ (((16 * w + 64 * s + rl) >> 2) & 3);
            const bool ok = b_off[s] != kOob16 && k0 + part * 8 < p.K;
            dma(db, st + (unsigned)(GA_T + 64 * s * GRS), ok ? b_off[s] + (unsigned)k0 * 2u : kOob16);
        }
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

    issue(0, 0);
    if (nk > 1) issue(1, 1);
    const int lswz = (li >> 2) & 3;
    const int a_row = (32 * MT * wm + li) * GRS, b_row = GA_T + (64 * wn + li) * GRS;
    int buf = 0;
#pragma unroll 1
    for (int k = 0; k < nk; ++k) {
        // stage k has landed once at most the next stage's GSA + GSB DMAs of this wave are still outstanding
        if (k + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(GSA + GSB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // every wave's slices of stage k are in LDS; everyone is done reading stage k - 1
        if (k + 2 < nk) issue(k + 2, buf == 0 ? 2 : buf - 1);       // (k + 2) % 3: the buffer stage k - 1 used
        const unsigned char* st = lds16 + buf * STG;
#pragma unroll
        for (int ks = 0; ks < GK / 16; ++ks) {
            const int co = ((2 * ks + lh) ^ lswz) * 16;
            uint4 fb[2], fa[MT];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) fb[nt] = *reinterpret_cast<const uint4*>(st + b_row + nt * 32 * GRS + co);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const uint4*>(st + a_row + mt * 32 * GRS + co);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = mfma_h16<kF16>(fb[nt], fa[mt], acc[mt][nt]);
        }
        buf = buf == 2 ? 0 : buf + 1;
    }
    if constexpr (EPI == 1) stats_tile<MT>(p, acc, m0, n0, wm, wn, li, lh);
    else if constexpr (EPI == 2) softmax_tile<MT>(p, acc, m0, n0, wm, wn, li, lh, lds16, t);
    else store_tile<MT>(p, acc, m0, n0, wm, wn, li, lh);
}

// ---- the wide kernel: one 64 MT x 256 tile per workgroup of EIGHT waves, one workgroup per CU (round 5) --------------------------------------
// The 256 x 128 / 192 x 128 kernel above is bound by the L2 -> CU operand stream on a big plain GEMM: attn.V of cs_attn at the C3 tile
// (36864 x 1024 x 9216) moves 20 KB per k-tile and workgroup = 8.85 GB per launch, 0.69 ms at the 12.9 TB/s the L2 delivers chip-wide
// (tools/ubench/l2_stream.hip) against 0.28 ms of MFMA issue: measured 0.79 ms.  A 192 x 256 tile moves 28 KB per k-tile for twice the
// MFMAs: 6.2 GB per launch, and 192 x 4 = 768 workgroups are exactly three rounds of 256 CUs.  Eight waves in a 2 x 4 grid, each the
// same MT x 2 MFMA tiles and the same k order per output element as above (bitwise the same result); stages of 28 slices of 1 KB + 4
// padding slices (every wave issues FOUR slices per stage: uniform counted waits), four stages, three k-tiles in flight.
constexpr int WN = 256;                                   // columns of the wide tile
constexpr int WNS = 4;                                    // stages
template <int MT>
__global__ __launch_bounds__(512) void gemm_h16_wide_kernel(Gemm16P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];      // [WNS] x { A [GM_][GRS], B [WN][GRS], 4 KB of padding slices }
    constexpr int GM_ = 64 * MT;
    constexpr int GA_T = GM_ * GRS;
    constexpr int NSL = (GM_ + WN) / 16;                   // real 16-row slices per stage (28 at MT = 3)
    static_assert(NSL <= 32 && GM_ % 16 == 0, "four slices per wave");
    constexpr int STG = 32 * 1024;                         // 32 slices
    const int bid = blockIdx.x;
    const int q8 = p.n_wg >> 3, r8 = p.n_wg & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int m0 = (lid / p.tiles_n) * GM_, n0 = (lid % p.tiles_n) * WN;

    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w >> 2, wn = w & 3, li = lane & 31, lh = lane >> 5;
    const int rl = lane >> 2, pc = lane & 3;
    // slice i = w + 8 s of a stage: rows 16 i .. 16 i + 15 of [A | B | padding]
    unsigned off[4];
    bool is_b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i = w + 8 * s, row = 16 * i + rl;
        const int part = pc ^ ((row >> 2) & 3);
        if (i < GM_ / 16) {
            const int gm = m0 + row;
            off[s] = gm < p.M ? (unsigned)gm * (unsigned)p.lda * 2u + (unsigned)part * 16u : kOob16;
            is_b[s] = false;
        } else if (i < NSL) {
            const int gn = n0 + row - GM_;
            off[s] = gn < p.N ? (unsigned)gn * (unsigned)p.ldb * 2u + (unsigned)part * 16u : kOob16;
            is_b[s] = true;
        } else {
            off[s] = kOob16;                               // padding slice: zeros into the stage's last 4 KB
            is_b[s] = false;
        }
    }
    const int nk = (p.K + GK - 1) / GK;
    const i32x4 da = {(int)(unsigned)(size_t)p.A, (int)(((size_t)p.A >> 32) & 0xFFFFu), (int)p.a_bytes, 0x00020000};
    const i32x4 db = {(int)(unsigned)(size_t)p.B, (int)(((size_t)p.B >> 32) & 0xFFFFu), (int)p.b_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds16;
    auto dma = [&](const i32x4& desc, unsigned lds_dst, unsigned voff) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(desc) : "memory");
    };
    auto issue = [&](int kt, int buf) {
        const int k0 = kt * GK;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int i = w + 8 * s;
            const int part = pc ^ (((16 * i + rl) >> 2) & 3);
            const bool ok = off[s] != kOob16 && k0 + part * 8 < p.K;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * STG) + (unsigned)i * 1024u);
            dma(is_b[s] ? db : da, dst, ok ? off[s] + (unsigned)k0 * 2u : kOob16);
        }
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

#pragma unroll
    for (int k = 0; k < WNS - 1; ++k)
        if (k < nk) issue(k, k);
    const int lswz = (li >> 2) & 3;
    const int a_row = (32 * MT * wm + li) * GRS, b_row = GA_T + (64 * wn + li) * GRS;
    int buf = 0;
#pragma unroll 1
    for (int k = 0; k < nk; ++k) {
        // stage k has landed once at most the 4 slices each of the stages behind it that this wave has requested are outstanding
        const int behind = nk - 1 - k < WNS - 2 ? nk - 1 - k : WNS - 2;
        if (behind >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // every wave's slices of stage k are in LDS; everyone is done reading stage k - 1
        if (k + WNS - 1 < nk) issue(k + WNS - 1, (buf + WNS - 1) & (WNS - 1));         // the buffer stage k - 1 used
        const unsigned char* st = lds16 + buf * STG;
#pragma unroll
        for (int ks = 0; ks < GK / 16; ++ks) {
            const int co = ((2 * ks + lh) ^ lswz) * 16;
            uint4 fb[2], fa[MT];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) fb[nt] = *reinterpret_cast<const uint4*>(st + b_row + nt * 32 * GRS + co);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const uint4*>(st + a_row + mt * 32 * GRS + co);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = mfma_h16<kF16>(fb[nt], fa[mt], acc[mt][nt]);
        }
        buf = (buf + 1) & (WNS - 1);
    }
    // Plain fp32 output (what the launcher sends here): the tile leaves in WHOLE 128-B lines.  store_tile writes 16 B per lane for 32
    // different rows per instruction (lane = output row in the accumulator layout); here each wave turns its 32 MT x 32 column halves
    // around in 4 MT KB of the (now idle) stage buffers -- ds_write_b128 in the accumulator layout, chunk-swizzled, ds_read_b128 row-major
    // -- and a store instruction covers 8 rows x 128 B.
    if (!p.c_bf16 && !p.bias && p.alpha == 1.f && (p.ldc & 3) == 0) {
        __syncthreads();                                   // every wave is done with the last stage
        unsigned char* ot = lds16 + w * (32 * MT * 128);
        float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (nt) wave_lds_sync();                       // the buffer is rewritten: after the last read of the first half
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 32 * mt + li, n4 = 2 * q + lh;
                    *reinterpret_cast<float4*>(ot + r * 128 + ((n4 ^ (r & 7)) << 4)) =
                        make_float4(acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]);
                }
            wave_lds_sync();                               // lanes read what OTHER lanes of this wave wrote
            const int rr = lane >> 3, c = lane & 7;
#pragma unroll
            for (int j = 0; j < 4 * MT; ++j) {
                const int r = 8 * j + rr, m = m0 + 32 * MT * wm + r, n = n0 + 64 * wn + 32 * nt + 4 * c;
                const float4 v = *reinterpret_cast<const float4*>(ot + r * 128 + ((c ^ (r & 7)) << 4));
                if (m < p.M && n < p.N) *reinterpret_cast<float4*>(C + (size_t)m * p.ldc + n) = v;
            }
        }
        return;
    }
    // store_tile's column arithmetic is 64 wn + ...: the four wave columns of the wide tile need nothing else
    store_tile<MT>(p, acc, m0, n0, wm, wn, li, lh);
}
static_assert((WNS & (WNS - 1)) == 0, "stage index arithmetic");

// fp32 rows -> bf16 rows (first `cols` columns, cols % 4 == 0); pad columns [cols, ld_dst) are zeroed
__global__ void cast_rows_h16_kernel(const float* __restrict__ src, int ld_src, unsigned short* __restrict__ dst, int ld_dst,
                                      long rows, int cols) {
    const int c4n = ld_dst >> 2;
    const long n = rows * c4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4n;
        const int c = (int)(i - r * c4n) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < cols) v = *reinterpret_cast<const float4*>(src + r * ld_src + c);
        *reinterpret_cast<uint2*>(dst + r * ld_dst + c) =
            pack_h16x4<kF16>(v.x, v.y, v.z, v.w);
    }
}

// row softmax of fp32 logits S [rows][ld] (first L columns) -> bf16 probabilities P [rows][ldp]; pad columns zeroed
__global__ __launch_bounds__(256) void softmax_rows_h16_kernel(const float* __restrict__ S, long rows, int L, int ld,
                                                                unsigned short* __restrict__ P, int ldp) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* s = reinterpret_cast<const float4*>(S + (size_t)row * ld);
    const int n4 = ldp >> 2;
    float m = -INFINITY;
    for (int t = lane; t < n4; t += 64) {
        const int c = 4 * t;
        if (c >= L) break;
        const float4 v = s[t];
        m = fmaxf(m, v.x);
        if (c + 1 < L) m = fmaxf(m, v.y);
        if (c + 2 < L) m = fmaxf(m, v.z);
        if (c + 3 < L) m = fmaxf(m, v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int t = lane; t < n4; t += 64) {
        const int c = 4 * t;
        if (c >= L) break;
        const float4 v = s[t];
        sum += expf(v.x - m);
        if (c + 1 < L) sum += expf(v.y - m);
        if (c + 2 < L) sum += expf(v.z - m);
        if (c + 3 < L) sum += expf(v.w - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    uint2* po = reinterpret_cast<uint2*>(P + (size_t)row * ldp);
    for (int t = lane; t < n4; t += 64) {
        const int c = 4 * t;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < L) {
            const float4 x = s[t];
            v.x = expf(x.x - m) / sum;
            v.y = c + 1 < L ? expf(x.y - m) / sum : 0.f;
            v.z = c + 2 < L ? expf(x.z - m) / sum : 0.f;
            v.w = c + 3 < L ? expf(x.w - m) / sum : 0.f;
        }
        po[t] = pack_h16x4<kF16>(v.x, v.y, v.z, v.w);
    }
}

// Row tile of a launch: 256 rows (MT = 4) unless 192-row tiles (MT = 3) cut the number of chip-wide rounds enough -- two workgroups
// per CU = 512 tiles per round, and a launch costs (rounds x rows per tile): the 192x192 tile's P.V' (M = 36864, N = 1024) is 1152
// tiles of 256 rows = 2.25 rounds, paid as 3 x 256, against 1536 tiles of 192 rows = 3 rounds x 192.
static int pick_mt(int M, int tiles_n) {
    const long c4 = (long)ceil_div((long)ceil_div(M, 256) * tiles_n, 512) * 4;
    const long c3 = (long)ceil_div((long)ceil_div(M, 192) * tiles_n, 512) * 3;
    return c3 * 10 < c4 * 9 ? 3 : 4;
}
template <int EPI>
static int launch_gemm16(Gemm16P& p, int mt, hipStream_t s) {
    p.n_wg = ceil_div(p.M, 64 * mt) * p.tiles_n;
    if (mt == 3) {
        CIAOSR_BIG_LDS((gemm_h16_kernel<EPI, 3>), kGemm16Lds);
        hipLaunchKernelGGL((gemm_h16_kernel<EPI, 3>), dim3(p.n_wg), dim3(256), kGemm16Lds, s, p);
    } else {
        CIAOSR_BIG_LDS((gemm_h16_kernel<EPI, 4>), kGemm16Lds);
        hipLaunchKernelGGL((gemm_h16_kernel<EPI, 4>), dim3(p.n_wg), dim3(256), kGemm16Lds, s, p);
    }
    return CIAOSR_OK;
}

int gemm_h16_nt(const unsigned short* A, int lda, const unsigned short* B, int ldb, void* C, int ldc, bool c_bf16, int M, int N,
                 int K, float alpha, hipStream_t s, const char* tag) {
    if (M <= 0 || N <= 0) return CIAOSR_OK;
    CIAOSR_CHECK_ARG(A && B && C && K > 0 && (K & 7) == 0);
    CIAOSR_CHECK_ARG((lda & 7) == 0 && (ldb & 7) == 0 && (ldc & 3) == 0 && aligned16(A) && aligned16(B) && aligned16(C));
    if (((size_t)(M - 1) * lda + K) * 2 >= 0xFFFFFF00ull && M > GM) {       // A past one buffer descriptor (4 GiB): independent row blocks
        const int half = (M / 2 + GM - 1) / GM * GM;
        const size_t c_row = (size_t)ldc * (c_bf16 ? 2 : 4);
        const int rc = gemm_h16_nt(A, lda, B, ldb, C, ldc, c_bf16, half, N, K, alpha, s, tag);
        if (rc != CIAOSR_OK) return rc;
        return gemm_h16_nt(A + (size_t)half * lda, lda, B, ldb, static_cast<char*>(C) + (size_t)half * c_row, ldc, c_bf16, M - half, N, K, alpha, s, tag);
    }
    Gemm16P p;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc; p.c_bf16 = c_bf16 ? 1 : 0;
    p.M = M; p.N = N; p.K = K; p.alpha = alpha;
    p.bias = nullptr; p.res = nullptr; p.ldres = 0; p.C2 = nullptr; p.ldc2 = 0; p.C16 = nullptr; p.ldc16 = 0;
    p.part = nullptr; p.n_part = 0; p.stats = nullptr; p.npad = 0;
    const size_t ab = ((size_t)(M - 1) * lda + K) * 2, bb = ((size_t)(N - 1) * ldb + K) * 2;
    CIAOSR_CHECK_ARG(ab < 0xFFFFFF00ull && bb < 0xFFFFFF00ull);
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    p.tiles_n = ceil_div(N, GN);
    ProfScope prof(tag ? tag : "gemm" CIAOSR_H16_SUFFIX, s);
    // big plain GEMMs (the operand stream from L2 bounds the narrow tile): 192 x 256 tiles, eight waves, where they fill the chip
    const long wide_tiles = (long)ceil_div(M, 192) * ceil_div(N, WN);
    static const bool narrow_only = getenv("CIAOSR_GEMM16_NARROW") != nullptr;          // developer A/B switch
    // (deep K only: at K = 576 -- the logit table's slices -- one workgroup per CU hides its 18 k-tiles' prologue worse than two: 45 vs 39 us)
    if (N % WN == 0 && K >= 2048 && wide_tiles >= 256 && !narrow_only) {
        p.tiles_n = N / WN;
        p.n_wg = (int)wide_tiles;
        const size_t lds = (size_t)WNS * 32 * 1024;
        CIAOSR_BIG_LDS((gemm_h16_wide_kernel<3>), lds);
        hipLaunchKernelGGL((gemm_h16_wide_kernel<3>), dim3(p.n_wg), dim3(512), lds, s, p);
        return launch_status("gemm_wide" CIAOSR_H16_SUFFIX);
    }
    const int rc = launch_gemm16<0>(p, pick_mt(M, p.tiles_n), s);
    if (rc != CIAOSR_OK) return rc;
    return launch_status("gemm" CIAOSR_H16_SUFFIX);
}

// One Linear of the staged 16-bit MLP (mlp_refiner.py:87-102): C = [relu](A . W^T + bias), A and W 16-bit, C fp32 or 16-bit.
// N a multiple of 4, K a multiple of 8 (the caller pads both operands with zero columns).
int linear_h16(const unsigned short* A, int lda, const unsigned short* W16, int ldw, const float* bias, bool relu, void* C, int ldc, bool c_16bit,
               int M, int N, int K, hipStream_t s, const char* tag) {
    CIAOSR_CHECK_ARG(A && W16 && bias && C && M > 0 && N > 0 && (N & 3) == 0 && K > 0 && (K & 7) == 0);
    CIAOSR_CHECK_ARG((lda & 7) == 0 && (ldw & 7) == 0 && (ldc & 3) == 0 && aligned16(A) && aligned16(W16) && aligned16(bias) && (((size_t)C) & 7) == 0);
    Gemm16P p;
    p.A = A; p.lda = lda; p.B = W16; p.ldb = ldw; p.C = C; p.ldc = ldc; p.c_bf16 = c_16bit ? 1 : 0;
    p.M = M; p.N = N; p.K = K; p.alpha = 1.f;
    p.bias = bias; p.res = nullptr; p.ldres = 0; p.C2 = nullptr; p.ldc2 = 0; p.C16 = nullptr; p.ldc16 = 0;
    p.part = nullptr; p.n_part = 0; p.stats = nullptr; p.npad = 0;
    p.relu = relu ? 1 : 0;
    const size_t ab = ((size_t)(M - 1) * lda + K) * 2, bb = ((size_t)(N - 1) * ldw + K) * 2;
    CIAOSR_CHECK_ARG(ab < 0xFFFFFF00ull && bb < 0xFFFFFF00ull);
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    p.tiles_n = ceil_div(N, GN);
    ProfScope prof(tag, s);
    const int rc = launch_gemm16<0>(p, pick_mt(M, p.tiles_n), s);
    if (rc != CIAOSR_OK) return rc;
    return launch_status("linear" CIAOSR_H16_SUFFIX);
}

int cast_rows_h16(const float* src, int ld_src, unsigned short* dst, int ld_dst, long rows, int cols, hipStream_t s) {
    CIAOSR_CHECK_ARG(src && dst && (ld_src & 3) == 0 && (ld_dst & 3) == 0 && (cols & 3) == 0 && cols <= ld_dst);
    ProfScope prof("cast_rows" CIAOSR_H16_SUFFIX, s);
    const long n = rows * (ld_dst >> 2);
    int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL(cast_rows_h16_kernel, dim3(grid > 8192 ? 8192 : grid), dim3(256), 0, s, src, ld_src, dst, ld_dst, rows, cols);
    return launch_status("cast_rows" CIAOSR_H16_SUFFIX);
}

int softmax_rows_h16(const float* S, long rows, int L, int ld, unsigned short* P, int ldp, hipStream_t s) {
    CIAOSR_CHECK_ARG(S && P && (ld & 3) == 0 && (ldp & 3) == 0 && ldp <= ld && L <= ldp);
    ProfScope prof("softmax_rows", s);
    if (softmax_rows_reg_h16(S, rows, L, ld, P, ldp, kF16, s)) return launch_status("softmax_rows_reg" CIAOSR_H16_SUFFIX);
    hipLaunchKernelGGL(softmax_rows_h16_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, S, rows, L, ld, P, ldp);
    return launch_status("softmax_rows" CIAOSR_H16_SUFFIX);
}

// ---- skinny 1x1 convolution: N = 64 output channels, K <= 576 -------------------------------------------------------------------
// The 256 x 128 GEMM tile above is latency-bound on this shape (M = 36864, N = 64, K = 576: 144 workgroups, two 24-KB k-tiles in
// flight each: 37 us against ~15 us of memory time).  Here the WEIGHTS are the resident operand -- [64][K] 16-bit in LDS (74 KB,
// two workgroups per CU) -- and every wave streams one 32-row tile of A straight from memory into MFMA operand registers: all
// K / 16 fragment loads of the tile (16 B per lane each) are issued before anything waits, so a wave has 36 loads in flight instead
// of 6.  Epilogue as conv1x1_h16 (bias, fp32 residual, fp32 + fp32 + 16-bit outputs).
constexpr int SK_KS = 36;                      // k16-steps held in registers (K <= 576)
constexpr int SK_N = 64;
struct Skinny16P {
    const unsigned short* A; int lda; unsigned a_bytes;
    const unsigned short* W; int ldw;          // [64][ldw]
    const float* bias;
    const float* res; int ldres;
    float* out; int ldo;
    float* out2; int ldo2;
    unsigned short* out16; int ldo16;
    int M, K, nks, wpitch;                     // wpitch: LDS row pitch in bytes
};

__global__ __launch_bounds__(256, 2) void conv1x1_skinny_h16_kernel(Skinny16P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];      // [64][wpitch]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
    const int rt = blockIdx.x * 4 + w;                                      // 32-row tile of this wave
    const int row = rt * 32 + li;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.A), 0, p.a_bytes, 0x00020000);
    // the wave's whole A tile: fragment ks of lane (li, lh) = A[row][16 ks + 8 lh .. + 7]
    i32x4 af[SK_KS];
    const unsigned abase = row < p.M ? (unsigned)row * (unsigned)p.lda * 2u + (unsigned)lh * 16u : kOob16;
#pragma unroll
    for (int ks = 0; ks < SK_KS; ++ks)
        af[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, (abase != kOob16 && ks < p.nks) ? (int)(abase + (unsigned)ks * 32u) : (int)kOob16, 0, 0);
    // weights -> LDS (16-byte chunks; row pitch K * 2 + 16 B keeps the 32 rows of a ds_read_b128 on distinct bank quads)
    const int cpr = p.K >> 3;                                               // chunks per row
    for (int c = t; c < SK_N * cpr; c += 256) {
        const int r = c / cpr, k8 = c - r * cpr;
        *reinterpret_cast<uint4*>(wl + r * p.wpitch + k8 * 16) = *reinterpret_cast<const uint4*>(p.W + (size_t)r * p.ldw + k8 * 8);
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    const unsigned char* wrow = wl + li * p.wpitch + lh * 16;
#pragma unroll
    for (int ks = 0; ks < SK_KS; ++ks) {
        if (ks < p.nks) {
            const uint4 a = make_uint4((unsigned)af[ks].x, (unsigned)af[ks].y, (unsigned)af[ks].z, (unsigned)af[ks].w);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const uint4 wf = *reinterpret_cast<const uint4*>(wrow + nt * 32 * p.wpitch + ks * 32);
                acc[nt] = mfma_h16<kF16>(wf, a, acc[nt]);
            }
        }
    }
    if (row >= p.M) return;
    // every epilogue load first (bias, residual: 16 float4 in flight): written load -> store per column group, the stores' possible
    // aliasing with res / bias keeps hipcc from hoisting the next group's loads and the wave pays eight L2 / HBM round trips in a row --
    // three times its 72 MFMAs
    float4 bq[2][4], rq[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = 32 * nt + 8 * q + 4 * lh;
            bq[nt][q] = *reinterpret_cast<const float4*>(p.bias + n);
            rq[nt][q] = p.res ? *reinterpret_cast<const float4*>(p.res + (size_t)row * p.ldres + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = 32 * nt + 8 * q + 4 * lh;
            const float4 b = bq[nt][q];
            float v0 = acc[nt][4 * q] + b.x, v1 = acc[nt][4 * q + 1] + b.y, v2 = acc[nt][4 * q + 2] + b.z, v3 = acc[nt][4 * q + 3] + b.w;
            if (p.res) {
                const float4 r = rq[nt][q];
                v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
            }
            *reinterpret_cast<float4*>(p.out + (size_t)row * p.ldo + n) = make_float4(v0, v1, v2, v3);
            if (p.out2) *reinterpret_cast<float4*>(p.out2 + (size_t)row * p.ldo2 + n) = make_float4(v0, v1, v2, v3);
            if (p.out16) *reinterpret_cast<uint2*>(p.out16 + (size_t)row * p.ldo16 + n) = pack_h16x4<kF16>(v0, v1, v2, v3);
        }
}

// Round 5: the same kernel with the A tile fetched as WHOLE 128-B lines.  Above, a load instruction covers 32 rows x 32 B: the CU's vector
// L1 looks up 32 lines per instruction and uses a quarter of each (the pattern that bounded the first cut of the chained head kernel,
// DESIGN 4.3e).  Here instruction j of line L covers rows 8 j .. 8 j + 7 whole (8 lanes a row, 16 B each): a quarter of the tag lookups
// for the same bytes, all K / 64 x 4 loads of the tile still in flight at once (registers), and the MFMA operand layout comes out of a
// per-wave 4-KB LDS transpose per line (ds_write_b128 of whole rows, chunk-swizzled; ds_read_b128 as the decode kernel reads its Z
// lines).  Eight waves per workgroup share ONE copy of the weights (138 KB of LDS, one workgroup per CU).  K a multiple of 64.
__global__ __launch_bounds__(512) void conv1x1_lines_h16_kernel(Skinny16P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];      // [64][wpitch] weights, then [8 waves][2][4 KB] transpose buffers
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
    const int rt = blockIdx.x * 8 + w;                                      // 32-row tile of this wave
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.A), 0, p.a_bytes, 0x00020000);
    constexpr int NL = SK_KS / 4;                                           // 128-B lines of a row (9 at K = 576)
    const int nl = p.nks >> 2;
    i32x4 ld[NL][4];
    {
        const int rr = lane >> 3, c = lane & 7;
#pragma unroll
        for (int L = 0; L < NL; ++L)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = rt * 32 + 8 * j + rr;
                const bool ok = r < p.M && L < nl;
                ld[L][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ok ? (int)((unsigned)r * (unsigned)p.lda * 2u + (unsigned)L * 128u + (unsigned)c * 16u) : (int)kOob16, 0, 0);
            }
    }
    const int cpr = p.K >> 3;                                               // chunks per row
    for (int c = t; c < SK_N * cpr; c += 512) {
        const int r = c / cpr, k8 = c - r * cpr;
        *reinterpret_cast<uint4*>(wl + r * p.wpitch + k8 * 16) = *reinterpret_cast<const uint4*>(p.W + (size_t)r * p.ldw + k8 * 8);
    }
    __syncthreads();
    unsigned char* tb = wl + SK_N * p.wpitch + w * 8192;                    // this wave's two 4-KB line buffers: [32 rows][128 B], swizzled
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    const unsigned char* wrow = wl + li * p.wpitch + lh * 16;
    // writer: lane -> (row 8 j + lane / 8, chunk lane % 8) at position chunk ^ ((row >> 1) & 7); reader: lane (li, lh), k-step sl -> chunk 2 sl + lh
    const int wr_r = lane >> 3, wr_c = lane & 7;
#pragma unroll
    for (int L = 0; L < NL; ++L) {
        if (L < nl) {
            unsigned char* buf = tb + (L & 1) * 4096;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 8 * j + wr_r;
                *reinterpret_cast<i32x4*>(buf + r * 128 + ((wr_c ^ ((r >> 1) & 7)) << 4)) = ld[L][j];
            }
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const i32x4 av = *reinterpret_cast<const i32x4*>(buf + li * 128 + (((2 * sl + lh) ^ ((li >> 1) & 7)) << 4));
                const uint4 a = make_uint4((unsigned)av.x, (unsigned)av.y, (unsigned)av.z, (unsigned)av.w);
                const int ks = 4 * L + sl;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const uint4 wf = *reinterpret_cast<const uint4*>(wrow + nt * 32 * p.wpitch + ks * 32);
                    acc[nt] = mfma_h16<kF16>(wf, a, acc[nt]);
                }
            }
        }
    }
    // Epilogue through the same 8 KB of LDS: the accumulators (lane = row, 4 columns per quad) are written as a [32][64] fp32 tile, chunk-
    // swizzled, and read back row-major -- lane -> (row 4 j + lane / 16, columns 4 (lane % 16) ..) -- so that the residual loads and the
    // stores cover whole 256-B rows (4 rows per instruction) instead of 32 B of 32 different rows.  Same sums in the same order.
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n4 = 8 * nt + 2 * q + lh;
            *reinterpret_cast<float4*>(tb + li * 256 + ((n4 ^ (li & 15)) << 4)) =
                make_float4(acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]);
        }
    wave_lds_sync();                                       // lanes read what OTHER lanes of this wave wrote
    const int er = lane >> 4, ec = lane & 15;
    const float4 b = *reinterpret_cast<const float4*>(p.bias + 4 * ec);
    float4 rq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int gr = rt * 32 + 4 * j + er;
        rq[j] = (p.res && gr < p.M) ? *reinterpret_cast<const float4*>(p.res + (size_t)gr * p.ldres + 4 * ec) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + er, gr = rt * 32 + r;
        const float4 a = *reinterpret_cast<const float4*>(tb + r * 256 + ((ec ^ (r & 15)) << 4));
        float v0 = a.x + b.x, v1 = a.y + b.y, v2 = a.z + b.z, v3 = a.w + b.w;
        if (p.res) { v0 += rq[j].x; v1 += rq[j].y; v2 += rq[j].z; v3 += rq[j].w; }
        if (gr < p.M) {
            *reinterpret_cast<float4*>(p.out + (size_t)gr * p.ldo + 4 * ec) = make_float4(v0, v1, v2, v3);
            if (p.out2) *reinterpret_cast<float4*>(p.out2 + (size_t)gr * p.ldo2 + 4 * ec) = make_float4(v0, v1, v2, v3);
            if (p.out16) *reinterpret_cast<uint2*>(p.out16 + (size_t)gr * p.ldo16 + 4 * ec) = pack_h16x4<kF16>(v0, v1, v2, v3);
        }
    }
}

// 1x1 convolution of a 16-bit channels-last map (the RDB's local feature fusion, mmedit RDB.lff called from
// ciaosr_net.py:337): out[m][n] = sum_k A[m][k] W16[n][k] + bias[n] + res[m][n], N % 4 == 0, written as fp32 to `out` (and `out2`
// when given) and as 16-bit to `out16` when given (the next block's input group, which makes its cast launch unnecessary).
int conv1x1_h16(const unsigned short* A, int lda, const unsigned short* W16, int ldw, const float* bias, const float* res, int ldres,
                float* out, int ldo, float* out2, int ldo2, unsigned short* out16, int ldo16, int M, int N, int K, hipStream_t s,
                const char* tag) {
    CIAOSR_CHECK_ARG(A && W16 && bias && out && M > 0 && N > 0 && (N & 3) == 0 && K > 0 && (K & 7) == 0);
    CIAOSR_CHECK_ARG((lda & 7) == 0 && (ldw & 7) == 0 && (ldo & 3) == 0 && (ldo2 & 3) == 0 && (ldo16 & 3) == 0 && (ldres & 3) == 0);
    CIAOSR_CHECK_ARG(aligned16(A) && aligned16(W16) && aligned16(out) && aligned16(bias) && (!res || aligned16(res)) &&
                     (!out2 || aligned16(out2)) && (!out16 || aligned16(out16)));
    if (N == SK_N && (K & 15) == 0 && K <= SK_KS * 16 && ((size_t)(M - 1) * lda + K) * 2 < 0xFFFFFF00ull) {
        Skinny16P q;
        q.A = A; q.lda = lda; q.a_bytes = (unsigned)(((size_t)(M - 1) * lda + K) * 2);
        q.W = W16; q.ldw = ldw; q.bias = bias; q.res = res; q.ldres = ldres;
        q.out = out; q.ldo = ldo; q.out2 = out2; q.ldo2 = ldo2; q.out16 = out16; q.ldo16 = ldo16;
        q.M = M; q.K = K; q.nks = K >> 4; q.wpitch = K * 2 + 16;
        const size_t lds = (size_t)SK_N * q.wpitch;
        ProfScope prof(tag, s);
        static const bool quarter_lines = getenv("CIAOSR_CONV1X1_QUARTER_LINES") != nullptr;       // developer A/B switch: the round-3 kernel
        if ((K & 63) == 0 && !quarter_lines) {
            const size_t lds8 = lds + 8 * 8192;
            CIAOSR_BIG_LDS(conv1x1_lines_h16_kernel, lds8);
            hipLaunchKernelGGL(conv1x1_lines_h16_kernel, dim3(ceil_div(M, 256)), dim3(512), lds8, s, q);
            return launch_status("conv1x1_lines" CIAOSR_H16_SUFFIX);
        }
        CIAOSR_BIG_LDS(conv1x1_skinny_h16_kernel, lds);
        hipLaunchKernelGGL(conv1x1_skinny_h16_kernel, dim3(ceil_div(M, 128)), dim3(256), lds, s, q);
        return launch_status("conv1x1_skinny" CIAOSR_H16_SUFFIX);
    }
    Gemm16P p;
    p.A = A; p.lda = lda; p.B = W16; p.ldb = ldw; p.C = out; p.ldc = ldo; p.c_bf16 = 0;
    p.M = M; p.N = N; p.K = K; p.alpha = 1.f;
    p.bias = bias; p.res = res; p.ldres = ldres; p.C2 = out2; p.ldc2 = ldo2; p.C16 = out16; p.ldc16 = ldo16;
    p.part = nullptr; p.n_part = 0; p.stats = nullptr; p.npad = 0;
    const size_t ab = ((size_t)(M - 1) * lda + K) * 2, bb = ((size_t)(N - 1) * ldw + K) * 2;
    CIAOSR_CHECK_ARG(ab < 0xFFFFFF00ull && bb < 0xFFFFFF00ull);
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    p.tiles_n = ceil_div(N, GN);
    ProfScope prof(tag, s);
    const int rc = launch_gemm16<0>(p, pick_mt(M, p.tiles_n), s);
    if (rc != CIAOSR_OK) return rc;
    return launch_status("conv1x1" CIAOSR_H16_SUFFIX);
}

// P = row_softmax(alpha * A . B^T) as 16-bit rows [M][ldp] (pad columns [N, ldp) zeroed) WITHOUT materialising the logits:
// the contraction runs twice (K is short: the correlation scores of CrossScaleAttention have K = 9C/2 against N = L columns) --
// pass 1 leaves per-strip (max, sum) partials, a merge kernel turns them into (row max, 1 / row sum), pass 2 recomputes the same
// logits (same code, same order: bit-identical) and writes exp(v - max) / sum.  Replaces an fp32 logit matrix written once and
// read once plus a separate row-softmax kernel.  scratch: M * (2 * ceil(N / 128) + 1) float2.
__global__ void softmax_stats_merge_kernel(const float2* __restrict__ part, int n_part, long M, float2* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float2* pr = part + (size_t)row * n_part;
    float mx = -INFINITY;
    for (int i = lane; i < n_part; i += 64) mx = fmaxf(mx, pr[i].x);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int i = lane; i < n_part; i += 64) {
        const float2 v = pr[i];
        if (v.x != -INFINITY) sum += v.y * __expf(v.x - mx);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) stats[row] = make_float2(mx, 1.f / sum);
}

size_t softmax_gemm_scratch_floats(long M, int N) { return (size_t)M * (2 * (size_t)ceil_div(N, GN) + 1) * 2 + 64; }

int softmax_gemm_h16_nt(const unsigned short* A, int lda, const unsigned short* B, int ldb, unsigned short* P, int ldp, int M, int N, int K,
                        float alpha, float* scratch, size_t scratch_floats, hipStream_t s, const char* tag) {
    if (M <= 0 || N <= 0) return CIAOSR_OK;
    CIAOSR_CHECK_ARG(A && B && P && scratch && K > 0 && (K & 7) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && (ldp & 7) == 0 && ldp >= N);
    CIAOSR_CHECK_ARG(aligned16(A) && aligned16(B) && aligned16(P) && aligned16(scratch) && scratch_floats >= softmax_gemm_scratch_floats(M, N));
    Gemm16P p;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = nullptr; p.ldc = 0; p.c_bf16 = 0;
    p.M = M; p.N = N; p.K = K; p.alpha = alpha;
    p.bias = nullptr; p.res = nullptr; p.ldres = 0; p.C2 = nullptr; p.ldc2 = 0; p.C16 = P; p.ldc16 = ldp;
    const size_t ab = ((size_t)(M - 1) * lda + K) * 2, bb = ((size_t)(N - 1) * ldb + K) * 2;
    CIAOSR_CHECK_ARG(ab < 0xFFFFFF00ull && bb < 0xFFFFFF00ull);
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
    p.tiles_n = ceil_div(N, GN);
    // (Several consecutive column tiles per workgroup as one software pipeline measured the same time on the 192x192 tile's scores
    // -- M = 36864, N = 9216, K = 288: each 256 x 128 tile pulls 221 KB through L2 for 18.9 MFLOP -- and was dropped.)
    CIAOSR_CHECK_ARG(alpha > 0.f);                    // the pass-1 maximum is taken on the raw accumulators
    p.n_part = 2 * p.tiles_n;
    p.part = reinterpret_cast<float2*>(scratch);
    float2* stats = p.part + (size_t)M * p.n_part;
    p.stats = stats;
    p.npad = (int)round_up((size_t)N, 8) <= ldp ? (int)round_up((size_t)N, 8) : ldp;
    ProfScope prof(tag, s);
    const int mt = pick_mt(M, p.tiles_n);
    int rc = launch_gemm16<1>(p, mt, s);
    if (rc != CIAOSR_OK) return rc;
    hipLaunchKernelGGL(softmax_stats_merge_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, s, p.part, p.n_part, (long)M, stats);
    rc = launch_gemm16<2>(p, mt, s);
    if (rc != CIAOSR_OK) return rc;
    return launch_status("softmax_gemm" CIAOSR_H16_SUFFIX);
}

// rows row0 + r * row_stride (r < nrows) of a 16-bit matrix back to fp32 (exact): dst[r][0 .. cols), cols % 4 == 0
__global__ void rows_to_f32_h16_kernel(const unsigned short* __restrict__ src, long ld_src, long row0, long row_stride, int nrows, int cols,
                                       float* __restrict__ dst, int ld_dst) {
    const int c4n = cols >> 2;
    const long n = (long)nrows * c4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / c4n), c = (int)(i - (long)r * c4n) * 4;
        const uint2 v = *reinterpret_cast<const uint2*>(src + (row0 + r * row_stride) * ld_src + c);
        *reinterpret_cast<float4*>(dst + (size_t)r * ld_dst + c) =
            make_float4(h16_lo<kF16>(v.x), h16_hi<kF16>(v.x), h16_lo<kF16>(v.y), h16_hi<kF16>(v.y));
    }
}
int rows_to_f32_h16(const unsigned short* src, long ld_src, long row0, long row_stride, int nrows, int cols, float* dst, int ld_dst,
                    hipStream_t s) {
    CIAOSR_CHECK_ARG(src && dst && nrows > 0 && cols > 0 && (cols & 3) == 0 && (ld_src & 3) == 0 && (ld_dst & 3) == 0 && cols <= ld_dst);
    ProfScope prof("rows_to_f32" CIAOSR_H16_SUFFIX, s);
    const long n = (long)nrows * (cols >> 2);
    int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL(rows_to_f32_h16_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, s, src, ld_src, row0, row_stride, nrows, cols,
                       dst, ld_dst);
    return launch_status("rows_to_f32" CIAOSR_H16_SUFFIX);
}

// up to 16 fp32 matrices [rows][cols] (row stride cols) -> one 16-bit array [n][rows][cols], one launch
struct CastManyP { const float* src[16]; int n; long each; };
__global__ void cast_many_h16_kernel(CastManyP p, unsigned short* __restrict__ dst) {
    const long total = (long)p.n * p.each / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4;
        const int m = (int)(e / p.each);
        const float4 v = *reinterpret_cast<const float4*>(p.src[m] + (e - (long)m * p.each));
        *reinterpret_cast<uint2*>(dst + e) = pack_h16x4<kF16>(v.x, v.y, v.z, v.w);
    }
}
int cast_many_h16(const float* const* src, int n, int rows, int cols, unsigned short* dst, hipStream_t s) {
    CIAOSR_CHECK_ARG(src && dst && n >= 1 && n <= 16 && ((long)rows * cols) % 4 == 0);
    CastManyP p;
    for (int i = 0; i < 16; ++i) p.src[i] = i < n ? src[i] : nullptr;
    for (int i = 0; i < n; ++i) CIAOSR_CHECK_ARG(src[i] && aligned16(src[i]));
    p.n = n; p.each = (long)rows * cols;
    ProfScope prof("cast_weights" CIAOSR_H16_SUFFIX, s);
    const long total = (long)n * p.each / 4;
    hipLaunchKernelGGL(cast_many_h16_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, p, dst);
    return launch_status("cast_weights" CIAOSR_H16_SUFFIX);
}

}  // namespace CIAOSR_H16_NS
}  // namespace ciaosr
