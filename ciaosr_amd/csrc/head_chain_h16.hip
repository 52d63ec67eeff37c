// Weights-stationary, register-chained form of the 16-bit fused head kernel (round 5; precision modes "bf16" and "f16", compiled
// once per element type like the other *_h16.hip units).  Replaces head_kv_fused_h16_kernel on the default 16-bit route; the
// 128-row kernels of head_fused_h16.hip stay selectable (head_route bit CIAOSR_HEAD_NO_CHAIN) and are the fallback.
//
// Why.  The 128-row kernels stream every layer's weight fragments from L2 into registers per 128 rows and sit on the L2 -> CU
// delivery limit (21 B/clk per CU, tools/ubench/l2_stream.hip): MFMA-busy 0.46.  Here the weights go through LDS and the
// activations never leave the register file:
//   * ONE persistent workgroup of 8 waves per CU, two per SIMD.  A wave owns ONE row tile of 32 (query, key sample) rows for ALL
//     256 columns of a layer (CM = 1; the 4-wave / two-tiles-per-wave form measured 14 % slower: a lone wave per SIMD has nobody to
//     cover its epilogue VALU and DMA issue).
//   * Swapped MFMA operands (weights = A, activations = B).  The accumulator tile of output columns 32T .. 32T+31 of layer l is,
//     after relu + cvt_pk, exactly two B-operand fragments (k-steps 2T, 2T+1) of layer l + 1 -- lane (m, g) holds columns
//     16 s + 8 (e >> 2) + 4 g + (e & 3), and the weight fragments are packed with the same k permutation (pack_chain_kernel) -- so
//     hidden activations are chained register to register: no LDS round trip, no barrier between layers.
//   * The weight stream of a pass (3 + 3 hidden layers, the 9C+Cn-column output layer of imnet_v: 68 tiles of 16 KB at C = 64) is
//     DMA'd (buffer_load ... lds, 1 KB per wave instruction, issued one at a time behind MFMAs) into a ring of four 16-KB slots, three
//     slots ahead of its use; every wave brings an eighth of each slot and reads all of it.  One barrier per slot, in front of it a
//     COUNTED s_waitcnt vmcnt: vmcnt retires in order, so "at most N outstanding" means "everything older than the youngest N has
//     landed"; N = the vector-memory operations this wave is KNOWN to have issued behind the slot's pieces (every such operation of
//     the kernels is an asm statement; hipcc's own loads only make a wait conservative).  Too small a count waits longer, too large
//     a count reads a slot before it landed: each count below carries its derivation.
//   * Gathers (table rows, value rows, logit-table rows, imnet_q's Z rows) are staged as WHOLE 128-B lines of small per-tile windows
//     by LDS-DMA into per-wave stages (8 KB a wave), chunk-swizzled on the source side so that the ds_read_b128 lane groups are
//     conflict-free: a 16-B-per-lane gather costs the CU's vector L1 a tag lookup per (lane pair, line) and bounded the first cut.
//   * layer 0 of imnet_k / imnet_v comes from the hoisted tables (head.hip): the table row of the key pixel is loaded straight into
//     the accumulator layout as the C operand of ONE K = 16 MFMA that adds W1[:, tail] . (rel_y, rel_x, scale_y, scale_x), both
//     factors as hi + lo pairs (all four cross terms in the 16 k slots): the fp32 tail term of the old kernels to ~2^-17.
//   * rows are QUERY-major (row m = 4 q + j, 8 queries per row tile): the four key samples of a query are the four lanes of a DPP
//     quad, the 4-way softmax and z = sum_j a_j value_j . w_v,j reduce with quad_perm moves.
//   * queries are walked in 16 x 4 blocks of the HR grid when the caller says they form one (ciaosr_options_t.query_grid_w; a hint
//     for the traversal order only): the rows of a wave then gather from ~9 distinct LR pixels.
// The logit always comes from the logit table (head.hip `use_table`); a key outside the query's 3x3 neighbourhood (cannot happen for
// 0 < cell < 1) raises a device flag, and the 128-row kernel -- launched behind this one, gated on that flag -- redoes the launch.
// Reference: mmedited/models/backbones/sr_backbones/ciaosr_net.py:195-216 (query_rgb), components/refiners/mlp_refiner.py:87-102.
#include "h16_util.h"
#include "index_math.h"
#include "ops.h"

namespace ciaosr {
namespace CIAOSR_H16_NS {
namespace chain {

constexpr bool kF16 = CIAOSR_F16 != 0;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
#define LDS3 __attribute__((address_space(3)))
typedef const LDS3 unsigned char* lds_cptr;

#ifdef CIAOSR_PROBE      // developer probe build (make probe; tools/head_probe.py 192 f16c): cycle stamps of one workgroup's passes
__device__ unsigned long long g_cprobe[256 * 16];
#define CPROBE(slot) do { if (threadIdx.x == 0 && pass_i == 1) g_cprobe[(blockIdx.x & 255) * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define CPROBE(slot) do { } while (0)
#endif

#ifndef CIAOSR_CHAIN_ABL      // developer ablations (timing only; results are wrong): 1 no value-row loads, 2 no Z stores, 4 no v-out epilogue
#define CIAOSR_CHAIN_ABL 0    // arithmetic, 8 no DMA pieces in v-out, 32 no table / logit-table loads, 64 no hidden-layer epilogues, 128 no
                              // barriers, 256 no DMA pieces in the hidden layers, 512 constant biases
#endif
constexpr int kAbl = CIAOSR_CHAIN_ABL;
#ifndef CIAOSR_CHAIN_WAVES
#define CIAOSR_CHAIN_WAVES 8      // measured (C3 tile, f16): 8 waves 2.21 ms, 4 waves 2.51 ms -- the partner wave of a SIMD covers a wave's epilogues
#endif
constexpr int CNW = CIAOSR_CHAIN_WAVES;   // waves per workgroup: 4 (one per SIMD, two row tiles each) or 8 (two per SIMD, one row tile each)
constexpr int CM = 8 / CNW;               // row tiles per wave
#ifndef CIAOSR_CHAIN_ISSUERS
#define CIAOSR_CHAIN_ISSUERS 4    // round 6: the weight-stream pieces are issued by the first 4 waves only -- one per SIMD, the OLDER wave of its pair.
#endif                            // A piece costs its wave 60-180 cycles of issue in which it feeds no MFMA; with every wave issuing its share at the same
                                  // k-steps both waves of a SIMD sat in that issue together (measured on the dense kernel of this round, dense_h16.hip:
                                  // 5590 -> 4810 cycles per stage).  The younger wave of every SIMD is now pure MFMA + LDS reads.  8 = every wave (round 5).
constexpr int CIW = CIAOSR_CHAIN_ISSUERS < CNW ? CIAOSR_CHAIN_ISSUERS : CNW;      // issuing waves
constexpr int CPW = 16 / CIW;             // 1-KB DMA pieces of a slot an ISSUING wave issues (the other waves: none)
constexpr bool kSplitIssue = CIW != CNW;
// counted waits: `s_waitcnt vmcnt(N)` with N = the vector-memory operations this wave is known to have issued BEHIND the ones it waits for;
// the weight pieces among them exist in the issuing waves only (WITH), the other waves count without them (WITHOUT)
template <int WITH, int WITHOUT>
__device__ __forceinline__ void vm_wait(bool issuer) {
    if (!kSplitIssue || issuer) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(WITH) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(WITHOUT) : "memory");
}
constexpr int CROWS = 32 * CM * CNW;      // rows per workgroup pass (256)
constexpr int CQ = CROWS / 4;             // queries per pass (64)
constexpr int CSLOT = 16 * 1024;          // ring slot = 16 fragments: one tile, or the hi / the lo half of a pair tile
constexpr int CRING = 4;                  // slots of the weight ring: one being read, two in flight, one being requested
constexpr int CNSTAGE = 4;                // gather stages per row tile
constexpr int CWIN = 16;                  // key pixels of a row tile's gather window (4 x 4 LR pixels)
constexpr int CSTAGE = CWIN * 128;        // one staged line set: a 128-B line of each window pixel's row (2 KB)
constexpr int CTILE = 16 * 1024;          // one 32-column tile of a 256-deep layer: 16 fragments of 1 KB
constexpr int CTAIL = 8 * 1024;           // tail fragments of one chain (8 tiles x 1 KB)
constexpr unsigned kOobC = 0xFFFFFFF0u;

__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (kF16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack2(float a, float b) { return pack_h16x2<kF16>(a, b); }
__device__ __forceinline__ unsigned pack_relu2(float a, float b) { return pack_relu_h16x2<kF16>(a, b); }

__device__ __forceinline__ f32x4 cload4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0));
}

// k index of element e (0..7) of lane group g in k-step s of a chained layer (see the header comment)
__host__ __device__ constexpr int chain_k(int s, int g, int e) { return 16 * s + 8 * (e >> 2) + 4 * g + (e & 3); }

// ---- packing ----------------------------------------------------------------------------------------------------------------------
// W [N][ld] fp32, K = 256 -> tiles of 16 fragments [ks][lane][8 h16] with the chain's k permutation; PAIRS: each tile is followed by
// the same tile of the residuals w - h16(w).  Rows >= N are zero.
__global__ void pack_chain_kernel(const float* __restrict__ W, int ld, int N, int n_tiles, int pairs, uint4* __restrict__ out) {
    const long per_tile = (long)(pairs ? 2 : 1) * 16 * 64;
    const long total = (long)n_tiles * per_tile;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long f = idx >> 6;
        const int T = (int)(f / (pairs ? 32 : 16)), fi = (int)(f % (pairs ? 32 : 16));
        const int lo = fi >> 4, ks = fi & 15;
        const int n = 32 * T + (lane & 31), g = lane >> 5;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = n < N ? W[(size_t)n * ld + chain_k(ks, g, e)] : 0.f;
            if (lo) v[e] -= h16_lo<kF16>(to_h16<kF16>(v[e]));
        }
        out[idx] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    }
}
// the 4 tail columns W1[n][fan .. fan+3] of layer 0 as the A operand of the K = 16 tail MFMA: lane group 0 holds (hi, hi), group 1 (lo, lo)
__global__ void pack_tail_kernel(const float* __restrict__ W1, int ld, int fan, uint4* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 8 * 64) return;
    const int lane = idx & 63, T = idx >> 6;
    const int n = 32 * T + (lane & 31), g = lane >> 5;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = W1[(size_t)n * ld + fan + e];
        if (g) v[e] -= h16_lo<kF16>(to_h16<kF16>(v[e]));
    }
    const unsigned a = pack2(v[0], v[1]), b = pack2(v[2], v[3]);
    out[idx] = make_uint4(a, b, a, b);
}

struct ChainP {
    FusedKVP kv;                    // coord, cell, q0, nq, chunk, H, W, U, ldu, u_bytes, k.table, v.table, bias_*, softmax_scale, Z, ldz, G
    const unsigned char* blob;      // [tail_k 8 KB][tail_v 8 KB][stream: n_slots x 16 KB]
    unsigned blob_bytes;
    int n_slots;                    // slots of one pass
    int n_vout;                     // 32-column tiles of imnet_v's output layer
    int grid_w;                     // > 0: the nq queries are rows of a row-major grid with grid_w columns (traversal hint)
    int n_pass;
    int* flag;                      // set when a key leaves the query's 3x3 neighbourhood (the launch is then redone by the old kernel)
};

// one row tile's per-lane state (row m = lane & 31 = 4 q + j)
struct RowState {
    unsigned koff;                  // byte offset of the key pixel's row in the layer-0 tables (kpix * 1024)
    unsigned uoff;                  // byte offset of its U row
    unsigned goff;                  // byte offset of the logit-table row, kOobC = none
    int grow;                       // its row index, 0x7FFFFFFF = none
    unsigned zoff;                  // byte offset of the query's Z row, kOobC = query out of range
    u32x4 q4;                       // B operand of the tail MFMA: (rel_y, rel_x, scale_y, scale_x) as hi (elements 0-3) + lo (4-7)
    float attn;
};

template <bool PAIRS>
struct Geo {
    static constexpr int SLOTS_PER_TILE = PAIRS ? 2 : 1;
    static constexpr int PIECES_PER_TILE = CPW * SLOTS_PER_TILE;    // DMA pieces a wave issues per tile
    static constexpr int STEPS = 16 * SLOTS_PER_TILE;               // k-steps (fragments) per tile
};

// ---- the stream: ring of four 16-KB slots, DMA three slots ahead ----------------------------------------------------------------------
struct Stream {
    i32x4 desc;
    unsigned lds0;                  // LDS byte address of the ring
    unsigned voff;                  // this lane's offset inside a slot: (CPW w) KB + lane * 16
    bool iss;                       // this wave issues weight pieces (wave-uniform)
    unsigned src0;                  // byte offset of the stream inside the blob
    int n_slots;                    // per pass
    int total;                      // slots this workgroup consumes in the launch
    int cur;                        // next slot to consume
    int ridx;                       // its ring buffer (cur % 4)
    // prefetch target of the slot being consumed; pf_voff = voff, or an out-of-range offset past the launch's last slot (the DMA then
    // writes zeros into a buffer nobody reads again: every slot issues its pieces, the wait counts below never change)
    unsigned pf_dst, pf_src, pf_voff;

    __device__ __forceinline__ void dma(unsigned lds_dst, unsigned soff, unsigned vo) const {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" :: "v"(vo), "s"(lds_dst), "s"(desc), "s"(soff) : "memory");
    }
    __device__ __forceinline__ void issue_whole(int slot) const {       // prologue (slot < CRING - 1): all of this wave's pieces of a slot at once
        if (!iss) return;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)slot * CSLOT + (voff & ~1023u));
        const unsigned src = __builtin_amdgcn_readfirstlane(src0 + (unsigned)(slot % n_slots) * CSLOT);
#pragma unroll
        for (int i = 0; i < CPW; ++i) dma(dst + i * 1024u, src + i * 1024u, voff);
    }
    // in front of slot `cur`: its pieces have landed -- counted wait: this wave issued the pieces of slots cur + 1 and cur + 2 behind them,
    // so "at most 2 CPW vector-memory operations outstanding" covers them whatever else (gather DMAs, stores) is younger still; every wave
    // is past slot cur - 1 (barrier), whose buffer then receives slot cur + 3 piece by piece
    // EXTRA: further vector-memory operations this wave is KNOWN to have issued behind slot cur's pieces (the v-out units' value-row
    // fetches and Z stores): they may stay in flight too
    template <int EXTRA = 0>
    __device__ __forceinline__ lds_cptr begin_slot(lds_cptr ring) {
        vm_wait<(CRING - 2) * CPW + EXTRA, EXTRA>(iss);
        if (!(kAbl & 128)) __builtin_amdgcn_s_barrier();
        const int nxt = cur + CRING - 1;
        const int nidx = (ridx + CRING - 1) & (CRING - 1);            // the buffer slot cur - 1 used
        pf_voff = nxt < total ? voff : kOobC;
        pf_dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)nidx * CSLOT + (voff & ~1023u));
        pf_src = __builtin_amdgcn_readfirstlane(src0 + (unsigned)(nxt % n_slots) * CSLOT);
        lds_cptr s = ring + ridx * CSLOT;
        ++cur;
        ridx = (ridx + 1) & (CRING - 1);
        return s;
    }
    __device__ __forceinline__ void piece(int i) const {
        if (!iss) return;
        dma(pf_dst + (unsigned)i * 1024u, pf_src + (unsigned)i * 1024u, pf_voff);
    }
};
static_assert(CRING == 4, "ring index arithmetic");

// ---- epilogue pieces ----------------------------------------------------------------------------------------------------------------
// relu + convert accumulator registers 4 qd .. 4 qd + 3 of tile T of row tile mi into their two packed registers of `out`
__device__ __forceinline__ void finish_quad(int T, const f32x16& c, u32x4 (&out)[16], int qd) {
    unsigned o0 = pack_relu2(c[4 * qd + 0], c[4 * qd + 1]), o1 = pack_relu2(c[4 * qd + 2], c[4 * qd + 3]);
    asm volatile("" : "+v"(o0), "+v"(o1));      // computed HERE (hipcc otherwise sinks every epilogue behind the layer)
    u32x4& o = out[2 * T + (qd >> 1)];
    if (qd & 1) { o.z = o0; o.w = o1; } else { o.x = o0; o.y = o1; }
}
__device__ __forceinline__ void finish_tile(int T, const f32x16 (&c)[CM], u32x4 (&out)[CM][16]) {
#pragma unroll
    for (int mi = 0; mi < CM; ++mi)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) finish_quad(T, c[mi], out[mi], qd);
}

__device__ __forceinline__ f32x16 bias_frag(const LDS3 float* bias, int T, int lh) {
    f32x16 b;
    if (kAbl & 512) { for (int i = 0; i < 16; ++i) b[i] = 0.25f; return b; }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const LDS3 f32x4*)(bias + 32 * T + 8 * g + 4 * lh);
        b[4 * g] = v.x; b[4 * g + 1] = v.y; b[4 * g + 2] = v.z; b[4 * g + 3] = v.w;
    }
    return b;
}

// ---- the logit, accumulated under the k chain's third hidden layer --------------------------------------------------------------------
// logit = h . G[query pixel, key offset] + c with h = that layer's output (ciaosr_net.py:211 after the exact fold of imnet_k's output layer,
// head.hip).  The 32 columns of tile T are two k-steps of packed registers as soon as tile T's epilogue has run (under tile T + 1's MFMAs):
// their 4 x 16 B of the lane's G row are requested at the start of tile T and multiplied in behind tile T + 1's epilogue pieces -- the 64
// divergent gathers of a standalone logit phase (9-15 k cycles with nothing to cover them) become 4 per tile under 32 MFMAs.
#ifndef CIAOSR_CHAIN_FUSED_LOGIT
#define CIAOSR_CHAIN_FUSED_LOGIT 0      // measured slower (C3 tile, f16: k hidden 30 k -> 48 k cycles per pass): the 32-line gathers of 8 waves keep the
#endif                                  // CU's vector L1 busy for the whole layer and the weight DMAs queue behind them
constexpr bool kFusedLogit = CIAOSR_CHAIN_FUSED_LOGIT != 0;
struct LogitAcc {
    __amdgpu_buffer_rsrc_t rs_g;
    unsigned goff[CM];              // byte offset of the row's G row, kOobC = none
    f32x4 gq[2][CM][4];             // G columns of tile T's outputs, requested one tile ahead
    float sum[CM];
    int lh;
    bool on;                        // false: the functions below are no-ops (v chain)
    // the staged logit-table window (kernel body): instruction i of its eight
    i32x4 g_desc;
    unsigned g_vo;                  // byte offset of the window's row 0, kOobC = no window (the instructions still issue, out of range)
    unsigned g_row_bytes, g_lds, g_lane;
    __device__ __forceinline__ void window_dma(int i) const {
        // row g_row0 + i whole (64 lanes x 16 B), lane position l receiving source chunk l ^ 2 i; the row offset sits in the range-checked
        // VGPR offset: rows past the table's end read zeros
        const unsigned vo = g_vo == kOobC ? kOobC : g_vo + (unsigned)i * g_row_bytes + ((g_lane ^ (unsigned)(2 * i)) << 4);
        const unsigned dst = __builtin_amdgcn_readfirstlane(g_lds + (unsigned)i * 1024u);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(vo), "s"(dst), "s"(g_desc) : "memory");
    }
    __device__ __forceinline__ void request(int T) {
        if (!on) return;
#pragma unroll
        for (int mi = 0; mi < CM; ++mi)
#pragma unroll
            for (int i = 0; i < 4; ++i) {      // k-step s = 2 T + (i >> 1), half (i & 1): columns 16 s + 8 (i & 1) + 4 lh .. + 3
                const unsigned col = (unsigned)(16 * (2 * T + (i >> 1)) + 8 * (i & 1) + 4 * lh);
                gq[T & 1][mi][i] = (kAbl & 32) ? f32x4{0.1f, 0.2f, 0.3f, 0.4f} : cload4(rs_g, goff[mi] == kOobC ? kOobC : goff[mi] + col * 4u);
            }
    }
    __device__ __forceinline__ void add(int T, const u32x4 (&out)[CM][16]) {
        if (!on) return;
#pragma unroll
        for (int mi = 0; mi < CM; ++mi)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4& x = out[mi][2 * T + (i >> 1)];
                const unsigned x0 = (i & 1) ? x.z : x.x, x1 = (i & 1) ? x.w : x.y;
                const f32x4 g = gq[T & 1][mi][i];
                sum[mi] += h16_lo<kF16>(x0) * g.x + h16_hi<kF16>(x0) * g.y + h16_lo<kF16>(x1) * g.z + h16_hi<kF16>(x1) * g.w;
            }
    }
};

// ---- one tile of a 256-deep layer: acc[mi] = bias + W_tile . in[mi], the previous tile's epilogue in its shadow ----------------------
// tb: LDS address of the tile's fragments (+ lane * 16).  PT >= 0: `prev` is the accumulator of tile PT of the layer whose outputs go
// to `pout` (finished here, behind k-steps 2 ..).  Returns this tile's accumulators.
// LT >= 0: this is tile LT of the k chain's third hidden layer (see LogitAcc): request its G columns, multiply in tile LT - 1's.
// GI >= 0: tile GI (0..7) of the k chain's first hidden layer issues its weight pieces early (k-steps 1 ..) and window_dma(GI), one of the eight
// logit-table window DMAs, behind them (k-step 3): one 1-KB request per wave and tile instead of a chip-wide 16-MB burst in front of
// the chain (every workgroup starts its passes together; the burst took 7-8 k cycles to land and stalled the chain for 3 k of them)
template <bool PAIRS, int PT, int LT = -1, int XS0 = 0, int XS1 = 0, int GI = -1>
__device__ __forceinline__ void tile_mma(lds_cptr ring, int lane, const u32x4 (&in)[CM][16], const f32x16& c0, f32x16 (&acc)[CM], const f32x16 (&prev)[CM],
                                         u32x4 (&pout)[CM][16], Stream& st, LogitAcc* lg = nullptr) {
    constexpr int STEPS = Geo<PAIRS>::STEPS;
    u32x4 a[3];
    lds_cptr tb = ring;
#pragma unroll
    for (int f = 0; f < STEPS; ++f) {
        const int ks = f & 15;
        if (ks == 0) {                                  // a slot = 16 fragments
            tb = (f == 0 ? st.template begin_slot<XS0>(ring) : st.template begin_slot<XS1>(ring)) + lane * 16;      // XS: see Layers::tile
            asm volatile("" : "+v"(tb));
            a[f % 3] = *(const LDS3 u32x4*)(tb);
            a[(f + 1) % 3] = *(const LDS3 u32x4*)(tb + 1024);
        }
        if (ks + 2 < 16) a[(f + 2) % 3] = *(const LDS3 u32x4*)(tb + (ks + 2) * 1024);
#pragma unroll
        for (int mi = 0; mi < CM; ++mi) acc[mi] = mfma(a[f % 3], in[mi][ks], f == 0 ? c0 : acc[mi]);
        if (!(kAbl & 64) && PT >= 0 && f >= 2 && f < 2 + 4 * CM) {
            const int piece = f - 2;
            finish_quad(PT, prev[piece >> 2], pout[piece >> 2], piece & 3);
        }
        if (LT >= 0 && f == 1) lg->request(LT);
        if (LT >= 1 && f == 2 + 4 * CM) lg->add(LT - 1, pout);
        constexpr int P1 = 11;
        if (GI >= 0 && f < 16) {
            if (ks >= 1 && ks < 1 + CPW) st.piece(ks - 1);
            if (ks == 1 + CPW) lg->window_dma(GI);
        } else if (!(kAbl & 256) && ks >= P1 && ks < P1 + CPW) st.piece(ks - P1);   // this wave's share of the slot three ahead, one behind a k-step
        __builtin_amdgcn_sched_barrier(0);
    }
}

// three hidden layers of a chain: act0 -> act1 -> act0 -> act1.  On entry `prev` holds the accumulators of layer-0 tile 7 (its outputs
// belong to act0[.][14..15]); on exit `prev` holds those of the last layer's tile 7 (outputs: act1[.][14..15], NOT yet converted).
template <bool PAIRS>
struct Layers {
    using G = Geo<PAIRS>;
    // GX (k chain): window DMA t is issued in the first slot of tile t < 8, behind that slot's (early) weight pieces.  Slot k waits for its
    // pieces, issued in slot k - 3: the window DMAs of slots k - 3 .. k - 1 are younger and may stay in flight (extra(k) of them)
    static constexpr int extra(int k) {
        int n = 0;
        for (int j = k - 3; j <= k - 1; ++j)
            if (j >= 0 && j % G::SLOTS_PER_TILE == 0 && j / G::SLOTS_PER_TILE < 8) ++n;
        return n;
    }
    // BX > 0: the caller issued BX vector-memory operations right in front of the chain's first slot (the decode kernel's fetch of the next
    // pass's first Z line): younger than the pieces of the first three slots, whose waits let them stay in flight
    template <int L, int T, bool GX = false, int BX = 0>
    static __device__ __forceinline__ void tile(lds_cptr ring, Stream& st, u32x4 (&a0)[CM][16], u32x4 (&a1)[CM][16], const LDS3 float* bias,
                                                f32x16 (&acc)[2][CM], int lane, LogitAcc* lg) {
        constexpr int t_lin = 8 * L + T;                                   // tile index inside the chain's hidden stream
        const f32x16 c0 = bias_frag(bias + 256 * L, T, lane >> 5);
        constexpr int cur = t_lin & 1;
        constexpr int PT = (T + 7) & 7;                                    // previous tile (of the previous layer when T == 0)
        // outputs of the previous tile: layer L - 1's output array when T == 0 (= this layer's input), else this layer's output array
        constexpr int first_slot = t_lin * G::SLOTS_PER_TILE;
        constexpr int XS0 = GX ? extra(first_slot) : (first_slot < 3 ? BX : 0), XS1 = GX ? extra(first_slot + 1) : (first_slot + 1 < 3 ? BX : 0);
        constexpr int GI = (GX && t_lin < 8) ? t_lin : -1;
        if constexpr (L == 2) {             // in = a0, out = a1; the k chain accumulates its logit here (lg->on)
            if constexpr (T == 0) tile_mma<PAIRS, PT, 0>(ring, lane, a0, c0, acc[cur], acc[cur ^ 1], a0, st, lg);
            else tile_mma<PAIRS, PT, T>(ring, lane, a0, c0, acc[cur], acc[cur ^ 1], a1, st, lg);
        } else if constexpr ((L & 1) == 0) {       // in = a0, out = a1
            if constexpr (T == 0) tile_mma<PAIRS, PT, -1, XS0, XS1, GI>(ring, lane, a0, c0, acc[cur], acc[cur ^ 1], a0, st, lg);
            else tile_mma<PAIRS, PT, -1, XS0, XS1, GI>(ring, lane, a0, c0, acc[cur], acc[cur ^ 1], a1, st, lg);
        } else {                            // in = a1, out = a0
            if constexpr (T == 0) tile_mma<PAIRS, PT, -1, XS0, XS1>(ring, lane, a1, c0, acc[cur], acc[cur ^ 1], a1, st);
            else tile_mma<PAIRS, PT, -1, XS0, XS1>(ring, lane, a1, c0, acc[cur], acc[cur ^ 1], a0, st);
        }
    }
    template <int L, bool GX = false, int BX = 0>
    static __device__ __forceinline__ void layer(lds_cptr ring, Stream& st, u32x4 (&a0)[CM][16], u32x4 (&a1)[CM][16], const LDS3 float* bias,
                                                 f32x16 (&acc)[2][CM], int lane, LogitAcc* lg) {
        tile<L, 0, GX, BX>(ring, st, a0, a1, bias, acc, lane, lg); tile<L, 1, GX, BX>(ring, st, a0, a1, bias, acc, lane, lg);
        tile<L, 2, GX, BX>(ring, st, a0, a1, bias, acc, lane, lg); tile<L, 3, GX>(ring, st, a0, a1, bias, acc, lane, lg);
        tile<L, 4, GX>(ring, st, a0, a1, bias, acc, lane, lg); tile<L, 5, GX>(ring, st, a0, a1, bias, acc, lane, lg);
        tile<L, 6, GX>(ring, st, a0, a1, bias, acc, lane, lg); tile<L, 7, GX>(ring, st, a0, a1, bias, acc, lane, lg);
    }
};

// ---- staged gathers ------------------------------------------------------------------------------------------------------------------
// A divergent 16-B-per-lane gather costs the CU's vector L1 one tag lookup per (lane pair, 128-B line) and leaves 3/4 of every line
// unused: 64 such loads per chain and wave for the table rows, 160 for the value rows -- the probe showed them, not the MFMAs, bounding
// the kernel.  The rows of a row tile (8 queries x 4 key samples of a 4 x 2 block of the target grid) come from a handful of LR pixels, so
// each row tile keeps a WINDOW of 4 x 4 key pixels (origin = the smallest ky, kx of its rows) and fetches, per 128-B line index, that
// line of all 16 window rows with TWO LDS-DMA instructions (8 lanes per line: 16 whole-line requests instead of 128 quarter-used ones)
// into a 2-KB stage of LDS; every lane then reads its 4 x 16 B from the stage of ITS pixel.  Chunks are XOR-swizzled with the pixel
// index on the source side so that the reads of different pixels fall on different banks.  A key pixel outside its tile's window (the
// target grid coarser than ~2 x 2 LR pixels per 4 x 2 queries: scale < 1) raises the launch's fallback flag.
struct Window {
    unsigned t_off[2];              // per DMA instruction i: byte offset of (window pixel (lane >> 3) + 8 i, source chunk) in a table, kOobC outside the map
    unsigned u_off[2];              // ... in U
    unsigned rd;                    // LDS byte offset of this lane's pixel inside a stage, chunk swizzle folded in: (slot * 128) | ((slot & 7) << 4 ^ ...)
    unsigned slot;                  // window pixel of this lane's row
};

struct Stager {
    unsigned lds_stage;             // LDS byte address of this wave's stages: [CM][CNSTAGE][CSTAGE]
    // the 128-B line `line` of all 16 window rows of row tile mi into stage `buf` (two instructions; `off` = Window::t_off / u_off)
    __device__ __forceinline__ void fetch(const i32x4& d, const unsigned (&off)[2], unsigned line, int mi, int buf) const {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_stage + (unsigned)(mi * CNSTAGE + buf) * CSTAGE);
        // the line offset rides in the scalar offset (no per-line VGPR: hipcc kept all sixteen per-line offsets of the table rows alive from
        // the k chain's layer 0 to the v chain's); an out-of-range window pixel stays out of range (the check is on the VGPR offset)
        const unsigned so = __builtin_amdgcn_readfirstlane(line * 128u);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" :: "v"(off[i]), "s"(dst + i * 1024u), "s"(d), "s"(so) : "memory");
    }
};
// this lane's 16 B of source chunk `chunk` (0..7) of its pixel's line in stage (mi, buf)
__device__ __forceinline__ f32x4 stage_read(const LDS3 unsigned char* stages, const Window& wn, int mi, int buf, int chunk) {
    return *(const LDS3 f32x4*)(stages + (mi * CNSTAGE + buf) * CSTAGE + wn.slot * 128u + (((unsigned)chunk ^ (wn.slot & 7u)) << 4));
}

// layer 0 from the hoisted table: acc = T[key pixel][32 T ..] + W1[:, tail] . q4 (one K = 16 MFMA per row tile), relu, convert -> act0.
// The table rows arrive straight in the accumulator layout (4 x 16 B per tile and row); tile T + 1's rows are requested before tile T's
// MFMA.  The last tile's accumulators are left in acc[1] for the first hidden tile to finish (tile index 7 is odd).
__device__ __forceinline__ void build_rows(const LDS3 unsigned char* tail, const i32x4& d_t, const Stager& sg, const LDS3 unsigned char* stages,
                                           const Window (&wn)[CM], const RowState (&rs)[CM], u32x4 (&a0)[CM][16], f32x16 (&acc)[2][CM], int lane) {
    const int lh = lane >> 5;
    // line T of the window rows = columns 32 T .. 32 T + 31 of the table rows = tile T; CNSTAGE stages per row tile: tiles T + 1 .. T + D are in
    // flight while tile T is read (tile T + D goes into the stage tile T - 1 was read from).  Wait counts: 2 CM instructions per tile; tile
    // T's have landed once at most those of the tiles behind it are outstanding.
    constexpr int D = CNSTAGE - 1;
#pragma unroll
    for (int T = 0; T < D; ++T)
#pragma unroll
        for (int mi = 0; mi < CM; ++mi) sg.fetch(d_t, wn[mi].t_off, T, mi, T % CNSTAGE);
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        if (T + D < 8) {
#pragma unroll
            for (int mi = 0; mi < CM; ++mi) sg.fetch(d_t, wn[mi].t_off, T + D, mi, (T + D) % CNSTAGE);
        }
        {
            constexpr int kBehind[8] = {D, D, D, D, D < 3 ? D : 3, D < 2 ? D : 2, D < 1 ? D : 1, 0};
            switch (kBehind[T]) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * CM) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" :: "i"(4 * CM) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" :: "i"(6 * CM) : "memory"); break;
            }
        }
        f32x16 tr[CM];
#pragma unroll
        for (int mi = 0; mi < CM; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = stage_read(stages, wn[mi], mi, T % CNSTAGE, 2 * g + lh);
                tr[mi][4 * g] = v.x; tr[mi][4 * g + 1] = v.y; tr[mi][4 * g + 2] = v.z; tr[mi][4 * g + 3] = v.w;
            }
        const u32x4 a = *(const LDS3 u32x4*)(tail + T * 1024 + lane * 16);
#pragma unroll
        for (int mi = 0; mi < CM; ++mi) acc[T & 1][mi] = mfma(a, rs[mi].q4, tr[mi]);
        if (T > 0) finish_tile(T - 1, acc[(T - 1) & 1], a0);      // the previous tile's conversion under this tile's fetch
        __builtin_amdgcn_sched_barrier(0);
    }
}

// hi + lo halves of (rel_y, rel_x, scale_y, scale_x) as the B operand of the tail MFMA
__device__ __forceinline__ u32x4 q4_operand(float r0, float r1, float r2, float r3) {
    const unsigned h0 = pack2(r0, r1), h1 = pack2(r2, r3);
    const float l0 = r0 - h16_lo<kF16>(h0), l1 = r1 - h16_hi<kF16>(h0), l2 = r2 - h16_lo<kF16>(h1), l3 = r3 - h16_hi<kF16>(h1);
    return u32x4{h0, h1, pack2(l0, l1), pack2(l2, l3)};
}

template <bool PAIRS>
__global__ __launch_bounds__(64 * CNW) void head_kv_chain_kernel(ChainP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using G = Geo<PAIRS>;
    lds_cptr ring = (lds_cptr)smem_raw;
    LDS3 unsigned char* tails = (LDS3 unsigned char*)ring + CRING * CSLOT;           // [2][8 KB]
    LDS3 float* lbias = (LDS3 float*)(tails + 2 * CTAIL);                           // [3][256] k, [3][256] v, [n_vout * 32] v out
    const FusedKVP& kv = p.kv;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // ---- once per workgroup: tail fragments and biases into LDS, the first three slots of the stream on their way ----------------
    for (int i = t; i < 2 * CTAIL / 16; i += 64 * CNW) *(LDS3 u32x4*)(tails + i * 16) = reinterpret_cast<const u32x4*>(p.blob)[i];
    for (int i = t; i < 256; i += 64 * CNW) {
#pragma unroll
        for (int l = 0; l < 3; ++l) { lbias[256 * l + i] = kv.k.bias_hidden[l][i]; lbias[768 + 256 * l + i] = kv.v.bias_hidden[l][i]; }
    }
    for (int i = t; i < 32 * p.n_vout; i += 64 * CNW) lbias[1536 + i] = i < kv.v.n_out ? kv.v.bias_out[i] : 0.f;
    Stream st;
    st.desc = i32x4{(int)(unsigned)(size_t)p.blob, (int)(((size_t)p.blob >> 32) & 0xFFFFu), (int)p.blob_bytes, 0x00020000};
    st.lds0 = (unsigned)(size_t)(LDS3 unsigned char*)ring;
    st.iss = !kSplitIssue || w < CIW;
    st.voff = (unsigned)(CPW * (st.iss ? w : 0)) * 1024u + (unsigned)lane * 16u;
    st.src0 = 2 * CTAIL;
    st.n_slots = p.n_slots;
    const int my_passes = (p.n_pass - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    st.total = my_passes * p.n_slots;
    st.cur = 0;
    st.ridx = 0;
    st.pf_dst = 0; st.pf_src = 0; st.pf_voff = kOobC;
    // gather stages of this wave: [CM][CNSTAGE][2 KB] behind the bias table
    const LDS3 unsigned char* stages = (const LDS3 unsigned char*)(lbias + 1536 + 32 * (p.n_vout + 1)) + (size_t)w * (CM * CNSTAGE * CSTAGE);
    Stager sg;
    sg.lds_stage = __builtin_amdgcn_readfirstlane((unsigned)(size_t)stages);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < CRING - 1; ++s) st.issue_whole(s);   // (a launch has at least one pass = at least 50 slots)

    const unsigned t_bytes = (unsigned)((size_t)kv.H * kv.W * 1024);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kv.G), 0, kv.g_bytes, 0x00020000);
    auto desc_of = [](const void* ptr, unsigned bytes) { return i32x4{(int)(unsigned)(size_t)ptr, (int)(((size_t)ptr >> 32) & 0xFFFFu), (int)bytes, 0x00020000}; };
    const i32x4 d_tk = desc_of(kv.k.table, t_bytes), d_tv = desc_of(kv.v.table, t_bytes), d_u = desc_of(kv.U, kv.u_bytes);
    const i32x4 d_z = desc_of(kv.Z, (unsigned)((size_t)kv.nq * kv.ldz * 2));
    const i32x4 d_g = desc_of(kv.G, kv.g_bytes);
    const bool grid = p.grid_w > 0;
    const int grid_h = grid ? kv.nq / p.grid_w : 0;
    const int nbx = grid ? (p.grid_w + 15) >> 4 : 1;

    // the queries of a pass: where row tile mi's lane sits (ql = index inside the launch, ok = inside it) and its coordinates / cells.
    // (Issuing the loads of pass n + 1 in front of pass n's v-out loop only moved their wait: hipcc waits for its own loads with vmcnt(0).)
    struct QIn { int ql; bool ok; float cy, cx, c0y, c0x, cqy, cqx; };
    auto load_queries = [&](int pass, QIn (&qi)[CM]) {
#pragma unroll
        for (int mi = 0; mi < CM; ++mi) {
            const int ql8 = li >> 2;
            QIn q;
            if (grid) {
                const int bx = pass % nbx, by = pass / nbx;
                const int g8 = w * CM + mi;           // row tile of the pass: a 4 x 2 block of queries, the eight of them 4 wide and 2 high
                const int x = 16 * bx + 4 * (g8 & 3) + (ql8 & 3), y = 4 * by + 2 * (g8 >> 2) + (ql8 >> 2);
                q.ok = x < p.grid_w && y < grid_h;
                q.ql = y * p.grid_w + x;
            } else {
                q.ql = (pass * 8 + w * CM + mi) * 8 + ql8;
                q.ok = q.ql < kv.nq;
            }
            q.ok = q.ok && pass < p.n_pass;
            q.cy = q.cx = q.c0y = q.c0x = q.cqy = q.cqx = 0.f;
            if (q.ok) {
                const long qq = kv.q0 + q.ql;
                const long c0 = kv.chunk > 0 ? (qq / kv.chunk) * kv.chunk : 0;
                q.cy = kv.coord[2 * qq]; q.cx = kv.coord[2 * qq + 1];
                q.c0y = kv.cell[2 * c0]; q.c0x = kv.cell[2 * c0 + 1];
                q.cqy = kv.cell[2 * qq]; q.cqx = kv.cell[2 * qq + 1];
            }
            qi[mi] = q;
        }
    };
    QIn qin[CM];

    int pass_i = 0;
#pragma unroll 1
    for (int pass = blockIdx.x; pass < p.n_pass; pass += gridDim.x, ++pass_i) {
        CPROBE(0);
        load_queries(pass, qin);
        // ---- index math: row m = 4 q + j of row tile mi --------------------------------------------------------------------------
        RowState rs[CM];
        Window wn[CM];
        int bad = 0;
        int g_row0 = 0;
        bool g_wide = false;
#pragma unroll
        for (int mi = 0; mi < CM; ++mi) {
            const int j = li & 3;
            int ky = 0x7FFF, kx = 0x7FFF;
            const int ql = qin[mi].ql;
            const bool ok = qin[mi].ok;
            RowState r;
            r.koff = 0; r.uoff = 0; r.goff = kOobC; r.grow = 0x7FFFFFFF; r.zoff = kOobC; r.attn = 0.f;
            float t4[4] = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const float cy = qin[mi].cy, cx = qin[mi].cx;
                const KeySample s = key_sample(cy, cx, qin[mi].c0y, qin[mi].c0x, kv.H, kv.W, j, 2);
                const int kpix = s.ky * kv.W + s.kx;
                ky = s.ky; kx = s.kx;
                t4[0] = s.rel_y; t4[1] = s.rel_x;
                t4[2] = mul_rn(qin[mi].cqy, (float)kv.H);
                t4[3] = mul_rn(qin[mi].cqx, (float)kv.W);
                const int iy = nearest_index(cy, kv.H), ix = nearest_index(cx, kv.W);
                if (iy >= 0 && iy < kv.H && ix >= 0 && ix < kv.W) {
                    const int oy = s.ky - iy, ox = s.kx - ix;
                    if (oy >= -1 && oy <= 1 && ox >= -1 && ox <= 1) {
                        r.grow = (iy * kv.W + ix) * 9 + (oy + 1) * 3 + (ox + 1);
                        r.goff = (unsigned)r.grow * (unsigned)kv.ldg * 4u;
                    }
                    else bad = 1;
                }
                r.koff = (unsigned)kpix * 1024u;
                r.uoff = (unsigned)kpix * (unsigned)kv.ldu * 4u;
                r.zoff = (unsigned)ql * (unsigned)kv.ldz * 2u;
            }
            r.q4 = q4_operand(t4[0], t4[1], t4[2], t4[3]);
            rs[mi] = r;
            // gather window of the row tile: origin = the smallest key row / column of its 32 rows (rows of out-of-range queries stand aside)
            int y0 = ky, x0 = kx, g0 = r.grow;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) { y0 = min(y0, __shfl_xor(y0, d, 64)); x0 = min(x0, __shfl_xor(x0, d, 64)); g0 = min(g0, __shfl_xor(g0, d, 64)); }
            // logit-table rows of the row tile: a window of 8 consecutive rows from the smallest one (at an integer scale the 32 rows of a
            // 4 x 2 block of queries share the query pixel and use 6 of its 9 rows); wider spreads take the four-round path
            g_row0 = __builtin_amdgcn_readfirstlane(g0);
            g_wide = __builtin_amdgcn_ballot_w64(r.grow != 0x7FFFFFFF && r.grow - g0 >= 8) != 0;
            Window wv;
            const int dy = ky - y0, dx = kx - x0;
            wv.slot = 0;
            if (ok) {
                if (dy < 4 && dx < 4) wv.slot = (unsigned)(dy * 4 + dx);
                else bad = 1;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pw = (lane >> 3) + 8 * i, py = y0 + (pw >> 2), px = x0 + (pw & 3);      // window pixel this lane fetches for
                const unsigned chunk = (unsigned)((lane & 7) ^ (pw & 7));                        // ... and the source chunk that lands in ITS place
                const bool in = y0 != 0x7FFF && py < kv.H && px < kv.W;
                const unsigned pix = (unsigned)(py * kv.W + px);
                wv.t_off[i] = in ? pix * 1024u + chunk * 16u : kOobC;
                wv.u_off[i] = in ? pix * (unsigned)kv.ldu * 4u + chunk * 16u : kOobC;
            }
            wn[mi] = wv;
        }
        if (bad) atomicOr(p.flag, 1);

        u32x4 act0[CM][16], act1[CM][16];
        f32x16 acc[2][CM];

        // ================= phi_k =================================================================================================
        CPROBE(1);
        build_rows(tails, d_tk, sg, stages, wn, rs, act0, acc, lane);
        CPROBE(2);
        LogitAcc lg;
        lg.rs_g = rs_g; lg.lh = lh; lg.on = kFusedLogit;
        {
            // the logit-table window goes into the (now idle) gather stages under the first hidden tile, see tile_mma GI.  A wide row tile
            // issues the same eight instructions out of range (zeros): the counted waits do not depend on the path
            const bool fast = !g_wide && g_row0 != 0x7FFFFFFF && !(kAbl & 32);
            lg.g_desc = d_g;
            lg.g_vo = fast ? (unsigned)g_row0 * (unsigned)kv.ldg * 4u : kOobC;
            lg.g_lane = (unsigned)lane;
            lg.g_row_bytes = (unsigned)kv.ldg * 4u;
            lg.g_lds = sg.lds_stage;
        }
#pragma unroll
        for (int mi = 0; mi < CM; ++mi) { lg.goff[mi] = rs[mi].goff; lg.sum[mi] = 0.f; }
        Layers<PAIRS>::template layer<0, !kFusedLogit>(ring, st, act0, act1, lbias, acc, lane, &lg);
        Layers<PAIRS>::template layer<1, !kFusedLogit>(ring, st, act0, act1, lbias, acc, lane, &lg);
        Layers<PAIRS>::template layer<2>(ring, st, act0, act1, lbias, acc, lane, &lg);
        finish_tile(7, acc[1], act1);            // layer 2 (third hidden layer) writes act1; its tile 7 sits in acc[(16 + 7) & 1] = acc[1]
        CPROBE(3);
        lg.add(7, act1);
        float g_const = 0.f;
        if (!kFusedLogit) {
            // The logit as a phase of its own: h3 . G[row] with the 32 logit-table rows of the row tile staged through this wave's gather
            // stages (8 KB) by LDS-DMA, 256 B of every row per round (k-steps 4 r .. 4 r + 3), four rounds.  hipcc's form of the direct
            // gather (32 x 16 B per lane) kept two loads in flight: 9.8 k cycles.  Instruction i of a round fetches rows 4 i .. 4 i + 3
            // (16 lanes a row; lane position p receives source chunk p ^ (row & 15): the readers of a ds_read_b128 lane group then fall
            // on 16 different bank quads).
            static_assert(CM == 1 && CNSTAGE * CSTAGE == 8192, "8 KB of stages per wave");
            float asum = 0.f;
            if (!g_wide && g_row0 != 0x7FFFFFFF && !(kAbl & 32)) {
                // the window's rows landed under the hidden layers (the fourth slot's counted wait covered them): row slot sm, chunk c =
                // 4 s + 2 hi4 + lh of k-step s at position c ^ 2 sm
                const unsigned gb = rs[0].goff;
                g_const = gb != kOobC ? kv.G[(size_t)(gb >> 2) + 256] : 0.f;
                const unsigned sm = gb != kOobC ? (unsigned)(rs[0].grow - g_row0) : 0u;
                const LDS3 unsigned char* rq[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) rq[x] = stages + sm * 1024u + (unsigned)lh * 16u + ((((unsigned)x) ^ sm) << 5);
#pragma unroll
                for (int half = 0; half < 4; ++half) {
                    f32x4 gv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int s4 = 4 * half + (i >> 1), hi4 = i & 1;
                        gv[i] = *(const LDS3 f32x4*)(rq[hi4 | ((s4 & 3) << 1)] + (s4 >> 2) * 256);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int s4 = 4 * half + (i >> 1), hi4 = i & 1;
                        const unsigned x0 = hi4 ? act1[0][s4].z : act1[0][s4].x, x1 = hi4 ? act1[0][s4].w : act1[0][s4].y;
                        asum += h16_lo<kF16>(x0) * gv[i].x + h16_hi<kF16>(x0) * gv[i].y + h16_lo<kF16>(x1) * gv[i].z + h16_hi<kF16>(x1) * gv[i].w;
                    }
                }
                if (gb == kOobC) asum = 0.f;
            } else {
                // four rounds through the stages: 256 B of every row per round (k-steps 4 r .. 4 r + 3); instruction i of a round fetches
                // rows 4 i .. 4 i + 3, 16 lanes a row, lane position p receiving source chunk p ^ (row & 15)
                const unsigned gb = rs[0].goff;
                g_const = (gb != kOobC && !(kAbl & 32)) ? kv.G[(size_t)(gb >> 2) + 256] : 0.f;
                int lo = lane;                      // opaque copy: the per-lane addresses below are loop invariants hipcc would otherwise hoist out
                asm volatile("" : "+v"(lo));        // of the pass loop and keep in 24 registers for the whole kernel
                const int lio = lo & 31, lho = lo >> 5;
                const LDS3 unsigned char* rd0 = stages + lio * 256;
#pragma unroll 1
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int rr = 4 * i + (lo >> 4);
                        const unsigned g = (unsigned)__shfl((int)gb, rr, 64);
                        const unsigned vo = ((kAbl & 32) || g == kOobC) ? kOobC : g + ((unsigned)((lo & 15) ^ (rr & 15)) << 4) + (unsigned)r * 256u;
                        const unsigned dst = __builtin_amdgcn_readfirstlane(sg.lds_stage + (unsigned)i * 1024u);
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(vo), "s"(dst), "s"(d_g) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (r == 0) asm volatile("" :: "v"(g_const));          // requested in front of round 0, here: hipcc must not sink the load to its use
                    f32x4 gv[8];
#pragma unroll
                    for (int c2 = 0; c2 < 8; ++c2) gv[c2] = *(const LDS3 f32x4*)(rd0 + ((((unsigned)(2 * c2 + lho)) ^ (unsigned)(lio & 15)) << 4));
                    u32x4 hr[4];                // act1[0][4 r .. 4 r + 3] (r is a run-time index: the loop is rolled to keep the path small)
#pragma unroll
                    for (int q = 0; q < 4; ++q) hr[q] = r == 0 ? act1[0][q] : (r == 1 ? act1[0][4 + q] : (r == 2 ? act1[0][8 + q] : act1[0][12 + q]));
#pragma unroll
                    for (int c2 = 0; c2 < 8; ++c2) {
                        const int q = c2 >> 1, hi4 = c2 & 1;
                        const unsigned x0 = hi4 ? hr[q].z : hr[q].x, x1 = hi4 ? hr[q].w : hr[q].y;
                        asum += h16_lo<kF16>(x0) * gv[c2].x + h16_hi<kF16>(x0) * gv[c2].y + h16_lo<kF16>(x1) * gv[c2].z + h16_hi<kF16>(x1) * gv[c2].w;
                    }
                }
            }
            lg.sum[0] = asum;
        }
        // ---- logit (accumulated under layer 2, LogitAcc) + the table's constant term, 4-way softmax over the quad --------------------
#pragma unroll
        for (int mi = 0; mi < CM; ++mi) {
            const unsigned gb = rs[mi].goff;
            float a = lg.sum[mi];
            a += __shfl_xor(a, 32, 64);
            if (kFusedLogit) { if (gb != kOobC) a += kv.G[(size_t)(gb >> 2) + 256]; }
            else a += g_const;                       // (0 from the out-of-range load of a row without a table entry)
            const float lg = a / kv.softmax_scale;
            float m = fmaxf(lg, quad_xor1(lg));
            m = fmaxf(m, quad_xor2(m));
            const float e = expf(lg - m);
            float den = e + quad_xor1(e);
            den += quad_xor2(den);
            rs[mi].attn = e / den;
        }

        // ================= phi_v =================================================================================================
        CPROBE(4);
        build_rows(tails + CTAIL, d_tv, sg, stages, wn, rs, act0, acc, lane);
        CPROBE(5);
        lg.on = false;
        Layers<PAIRS>::template layer<0>(ring, st, act0, act1, lbias + 768, acc, lane, &lg);
        Layers<PAIRS>::template layer<1>(ring, st, act0, act1, lbias + 768, acc, lane, &lg);
        Layers<PAIRS>::template layer<2>(ring, st, act0, act1, lbias + 768, acc, lane, &lg);
        finish_tile(7, acc[1], act1);
        CPROBE(6);
        // ---- output layer fused with z = sum_j a_j value_j . (W h_j + b): one 32-column unit per tile of the stream ------------------
        {
            const int j = li & 3;
            // value rows: line u of the window rows (U columns 32 u .. 32 u + 31) through the gather stages; unit u's line is fetched during
            // unit u - 1 (behind its last read of unit u - 2's line, which shares the stage) and read during unit u + 1
            constexpr int PPT = G::PIECES_PER_TILE;
            auto fetch_vv = [&](int u) {
                if (kAbl & 1) return;
#pragma unroll
                for (int mi = 0; mi < CM; ++mi) sg.fetch(d_u, wn[mi].u_off, (unsigned)u, mi, u & (CNSTAGE - 1));
            };
            auto read_vv = [&](int u, int mi, int g) -> f32x4 {
                if (kAbl & 1) return f32x4{1.f, 1.f, 1.f, 1.f};
                return stage_read(stages + (u & (CNSTAGE - 1)) * CSTAGE, wn[mi], mi, 0, 2 * g + lh);
            };
            // epilogue of a unit in 8 + 2 pieces, each behind one k-step of the NEXT unit: piece (mi, g) reduces accumulator registers 4 g ..
            // 4 g + 3 of row tile mi over the quad (the four key samples of a query) and keeps them in the lane whose sample index is g
            float zk[CM][4] = {};
            // One asm block of 16 (g = 0) or 20 VALU instructions: t = (a . value) . acc for the 4 registers, the two quad butterflies as
            // v_add_f32_dpp, and for g > 0 a v_cndmask under the constant lane mask "sample index == g".  Every DPP source was written >= 3
            // instructions earlier (the 2-wait-state rule hipcc cannot apply inside an asm statement; its own form of this code carried an
            // s_nop per butterfly and an extra move); the accumulators are more than a k-step older than their MFMA.
            auto epi_piece = [&](int mi, int g, const f32x16& c, const f32x4& val) {
                if (kAbl & 4) { if (g == 0) { zk[mi][0] = c[0]; zk[mi][1] = c[5]; zk[mi][2] = c[10]; zk[mi][3] = c[15] + val[0]; } return; }
                float t0, t1, t2, t3;
#define CIAOSR_EPI_HEAD                                                                                                                \
                "v_mul_f32 %4, %8, %9\n\tv_mul_f32 %5, %8, %10\n\tv_mul_f32 %6, %8, %11\n\tv_mul_f32 %7, %8, %12\n\t"                             \
                "v_mul_f32 %4, %4, %13\n\tv_mul_f32 %5, %5, %14\n\tv_mul_f32 %6, %6, %15\n\tv_mul_f32 %7, %7, %16\n\t"                           \
                "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                                                  \
                "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                                                  \
                "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                                                  \
                "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                if (g == 0) {
                    asm volatile(CIAOSR_EPI_HEAD
                                 "v_add_f32_dpp %0, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %1, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %2, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %3, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                                 : "=&v"(zk[mi][0]), "=&v"(zk[mi][1]), "=&v"(zk[mi][2]), "=&v"(zk[mi][3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                                 : "v"(rs[mi].attn), "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
                } else {
                    const unsigned long long lanes_g = 0x1111111111111111ull << g;        // lanes whose key-sample index is g
                    asm volatile(CIAOSR_EPI_HEAD
                                 "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                                 "v_cndmask_b32 %0, %0, %4, %17\n\tv_cndmask_b32 %1, %1, %5, %17\n\tv_cndmask_b32 %2, %2, %6, %17\n\tv_cndmask_b32 %3, %3, %7, %17"
                                 : "+v"(zk[mi][0]), "+v"(zk[mi][1]), "+v"(zk[mi][2]), "+v"(zk[mi][3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                                 : "v"(rs[mi].attn), "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(c[4 * g]), "v"(c[4 * g + 1]),
                                   "v"(c[4 * g + 2]), "v"(c[4 * g + 3]), "s"(lanes_g));
                }
#undef CIAOSR_EPI_HEAD
            };
            auto epi_store = [&](int u, int mi) {
                const uint2 zb = pack_h16x4<kF16>(zk[mi][0], zk[mi][1], zk[mi][2], zk[mi][3]);
                const int d0 = 32 * u + 8 * j + 4 * lh;
                // the store is an asm statement like every other vector-memory instruction of this loop: the wait counts below count them all
                const unsigned zo = (rs[mi].zoff == kOobC || d0 >= kv.v.n_out || u < 0) ? kOobC : rs[mi].zoff + (unsigned)d0 * 2u;
                const unsigned long long zd = (unsigned long long)zb.x | ((unsigned long long)zb.y << 32);
                if (!(kAbl & 2)) asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" :: "v"(zd), "v"(zo), "s"(d_z) : "memory");
            };
            static_assert(CM == 1, "the pipelined v-out epilogue is written for one row tile per wave");
            // Vector-memory operations of a unit, in issue order: [k-step 0] the Z store of unit u - 2, [k-step 5] the two fetch instructions
            // of line u + VD, [k-steps 11, 12 of each slot] PPT weight pieces: OPU of them.  Line L is first read behind the last k-step of
            // unit L (value rows of accumulator registers 0-3 of the epilogue that runs under unit L + 1): behind its fetch this wave has
            // issued KW operations by then; lines 0 .. VD - 1 are fetched in front of the loop.
            constexpr int VD = 2, OPU = 3 + PPT;
            constexpr int KW = PPT + VD * OPU;
            constexpr int OPU0 = 3, KW0 = VD * OPU0;          // the same counts in a wave that issues no weight pieces
            auto wait_line = [&](int line) {        // uniform
                if (line >= VD) vm_wait<KW, KW0>(st.iss);
                else if (line == 0) vm_wait<2 * (VD - 1) + OPU, 2 * (VD - 1) + OPU0>(st.iss);
                else vm_wait<2 * OPU, 2 * OPU0>(st.iss);
            };
            static_assert(VD == 2 && VD + 2 <= CNSTAGE, "a line's stage is free again two units after its reads");
            auto epilogue = [&](int u, const f32x16 (&c)[CM]) {       // whole, for the last unit (its line has been waited for)
#pragma unroll
                for (int g = 0; g < 4; ++g) epi_piece(0, g, c[0], read_vv(u, 0, g));
            };
            // The epilogue of unit u - 1 under unit u, one accumulator register r = 4 g + i at a time in three stages a k-step apart, so that the
            // VALU work is spread evenly (about 5 instructions behind every MFMA; the first cut's 16-20-instruction blocks behind four of
            // the sixteen MFMAs ran in series with the MFMAs of BOTH waves of the SIMD) and every DPP source is a k-step old:
            //   A(r): t_r = (attn . value_r) . acc_r      B(r): t_r += quad-neighbour 1      C(r): t_r += quad-neighbour 2, kept by sample g's lane
            float tt[16];
            auto stage_a = [&](int r, const f32x16& c, const f32x4& val) {
                if (kAbl & 4) { tt[r] = c[r] + val[r & 3]; return; }
                asm volatile("v_mul_f32 %0, %1, %2\n\tv_mul_f32 %0, %0, %3" : "=&v"(tt[r]) : "v"(rs[0].attn), "v"(val[r & 3]), "v"(c[r]));
            };
            auto stage_b = [&](int r) {
                if (kAbl & 4) return;
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(tt[r]));
            };
            auto stage_c = [&](int r) {
                const int g = r >> 2, i = r & 3;
                if (kAbl & 4) { if (g == 0) zk[0][i] = tt[r]; return; }
                if (g == 0) {
                    asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(zk[0][i]) : "v"(tt[r]));
                } else {
                    const unsigned long long lanes_g = 0x1111111111111111ull << g;        // lanes whose key-sample index is g
                    asm volatile("v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32 %0, %0, %1, %2"
                                 : "+v"(zk[0][i]), "+v"(tt[r]) : "s"(lanes_g));
                }
            };
#pragma unroll
            for (int l = 0; l < VD; ++l) fetch_vv(l);
            const LDS3 float* bo = lbias + 1536;
            f32x4 va = f32x4{0.f, 0.f, 0.f, 0.f}, vb = va;       // value rows of register groups g = 0, 2 / g = 1, 3
            // two units per iteration: static accumulator buffers; the stream pads an odd unit count with a zero tile
            const int n_u2 = (p.n_vout + 1) >> 1;
#pragma unroll 1
            for (int u2 = 0; u2 < n_u2; ++u2) {
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    const int u = 2 * u2 + par;
                    const f32x16 c0 = bias_frag(bo, u, lh);
                    constexpr int STEPS = G::STEPS;
                    u32x4 a[3];
                    lds_cptr tb = ring;
                    const f32x16& cprev = acc[par ^ 1][0];
#pragma unroll
                    for (int f = 0; f < STEPS; ++f) {
                        const int ks = f & 15;
                        if (ks == 0) {
                            // behind slot cur's pieces (issued three slots ago): the pieces of slots cur + 1, cur + 2, the stores and fetches (3 a
                            // unit, k-steps 0 and 5 of its first slot) of the v-out units among the last two slots, the 2 VD fetches in front of
                            // the loop while the pieces date from the hidden layers
                            constexpr int PRE = 2 * VD;
                            lds_cptr sl;
                            if (!PAIRS) {
                                if (u == 0) sl = st.template begin_slot<PRE>(ring);
                                else if (u == 1) sl = st.template begin_slot<PRE + 3>(ring);
                                else if (u == 2) sl = st.template begin_slot<PRE + 6>(ring);
                                else sl = st.template begin_slot<6>(ring);
                            } else if (f == 0) {
                                if (u == 0) sl = st.template begin_slot<PRE>(ring);
                                else if (u == 1) sl = st.template begin_slot<PRE + 3>(ring);
                                else sl = st.template begin_slot<3>(ring);
                            } else {
                                if (u == 0) sl = st.template begin_slot<PRE + 3>(ring);
                                else sl = st.template begin_slot<3>(ring);
                            }
                            tb = sl + lane * 16;
                            asm volatile("" : "+v"(tb));
                            a[f % 3] = *(const LDS3 u32x4*)(tb);
                            a[(f + 1) % 3] = *(const LDS3 u32x4*)(tb + 1024);
                        }
                        if (ks + 2 < 16) a[(f + 2) % 3] = *(const LDS3 u32x4*)(tb + (ks + 2) * 1024);
                        acc[par][0] = mfma(a[f % 3], act1[0][ks], f == 0 ? c0 : acc[par][0]);
                        if (f == 0) epi_store(u - 2, 0);                                   // formed under unit u - 1
                        if (f == 2) vb = read_vv(u - 1, 0, 1);
                        if (f == 6) va = read_vv(u - 1, 0, 2);
                        if (f == 10) vb = read_vv(u - 1, 0, 3);
                        if (f < 16) {
                            // registers whose stage A sits at k-step x: x for x < 12, (12, 13) at 12, (14, 15) at 13
                            auto first = [](int x) { return x < 0 ? 0 : (x < 12 ? x : (x == 12 ? 12 : 14)); };
                            auto count = [](int x) { return x < 0 || x > 13 ? 0 : (x < 12 ? 1 : 2); };
                            if (count(f - 2) >= 1) stage_c(first(f - 2));
                            if (count(f - 2) == 2) stage_c(first(f - 2) + 1);
                            if (count(f - 1) >= 1) stage_b(first(f - 1));
                            if (count(f - 1) == 2) stage_b(first(f - 1) + 1);
                            if (count(f) >= 1) stage_a(first(f), cprev, ((first(f) >> 2) & 1) ? vb : va);
                            if (count(f) == 2) stage_a(first(f) + 1, cprev, (((first(f) + 1) >> 2) & 1) ? vb : va);
                        }
                        if (f == 5) fetch_vv(u + VD);
                        if (!(kAbl & 8) && ks >= 11 && ks < 11 + CPW) st.piece(ks - 11);
                        if (f == STEPS - 1) { wait_line(u); va = read_vv(u, 0, 0); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // drain: Z of the last-but-one unit run; an even unit count leaves the last unit's epilogue to do (an odd count's pad unit ran it)
            const int u_end = 2 * n_u2;
            epi_store(u_end - 2, 0);
            if (!(p.n_vout & 1)) {
                epilogue(u_end - 1, acc[1]);
                epi_store(u_end - 1, 0);
            }
        }
        CPROBE(7);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA of this workgroup may land after it has gone
}


// =================================================================================================================================
// imnet_q in the same form: rgb = W5 relu(W4 relu(W3 relu(W2 relu(W1 z + b1) ..))) + b5 + bilinear(x_lr).  Rows are queries (256 a pass, 32 a
// wave), the weight stream of a pass is the input layer (k-step major: the eight 32-column tiles of a k-step back to back, so that a Z
// fragment is read once and feeds eight independent accumulators), the three hidden layers (Layers, as above) and the 3-row output layer
// as ONE tile of hi + lo halves (the fp32 weights of the old kernels' VALU tail to 2^-22).  The Z rows of a wave are staged like the
// value rows above: whole 128-B lines (4 k-steps) of its 32 rows by four LDS-DMA instructions into a 4-KB stage, two stages; the next
// pass's first line is fetched under the hidden layers.  Reference: ciaosr_net.py:107-108, 221 (imnet_q + the bilinear residual).
struct DecodeChainP {
    FusedQP q;
    const unsigned char* blob;      // [stream: n_slots x 16 KB]
    unsigned blob_bytes;
    int n_slots;                    // per pass
    int nline;                      // 128-B lines of a Z row (Dv / 64), even
    int n_pass;                     // passes of 256 queries
};

// input-layer fragments, k-step major: [ks][hi: 8 tiles | lo: 8 tiles (pairs)][lane][8 h16], lane (n, g) holding W[32 T + n][16 ks + 8 g + e]
__global__ void pack_decode_in_kernel(const float* __restrict__ W, int ld, int nks, int pairs, uint4* __restrict__ out) {
    const int per_ks = pairs ? 16 : 8;
    const long total = (long)nks * per_ks * 64;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long f = idx >> 6;
        const int ks = (int)(f / per_ks), r = (int)(f % per_ks), lo = r >> 3, T = r & 7;
        const int n = 32 * T + (lane & 31), g = lane >> 5;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = W[(size_t)n * ld + 16 * ks + 8 * g + e];
            if (lo) v[e] -= h16_lo<kF16>(to_h16<kF16>(v[e]));
        }
        out[idx] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    }
}

// one slot (16 fragments) of the input layer: single weights = k-steps (2 n, 2 n + 1) x 8 tiles, pairs = one k-step's hi and lo x 8 tiles
template <bool PAIRS, int EXTRA>
__device__ __forceinline__ void decode_in_slot(lds_cptr ring, Stream& st, int lane, const u32x4& b0, const u32x4& b1, f32x16 (&acc8)[8]) {
    lds_cptr tb = st.template begin_slot<EXTRA>(ring) + lane * 16;
    asm volatile("" : "+v"(tb));
    u32x4 a[3];
    a[0] = *(const LDS3 u32x4*)(tb);
    a[1] = *(const LDS3 u32x4*)(tb + 1024);
#pragma unroll
    for (int f = 0; f < 16; ++f) {
        if (f + 2 < 16) a[(f + 2) % 3] = *(const LDS3 u32x4*)(tb + (f + 2) * 1024);
        acc8[f & 7] = mfma(a[f % 3], (PAIRS || f < 8) ? b0 : b1, acc8[f & 7]);
        if (f >= 11 && f < 11 + CPW) st.piece(f - 11);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool PAIRS>
__global__ __launch_bounds__(64 * CNW) void head_decode_chain_kernel(DecodeChainP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    static_assert(CM == 1 && CNW == 8, "eight waves, one 32-query row tile each");
    lds_cptr ring = (lds_cptr)smem_raw;
    LDS3 float* lbias = (LDS3 float*)((LDS3 unsigned char*)ring + CRING * CSLOT);       // [256] in, [3][256] hidden, [32] out
    const FusedQP& q = p.q;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    for (int i = t; i < 256; i += 64 * CNW) {
        lbias[i] = q.bias_in[i];
#pragma unroll
        for (int l = 0; l < 3; ++l) lbias[256 * (l + 1) + i] = q.bias_hidden[l][i];
    }
    if (t < 32) lbias[1024 + t] = t < 3 ? q.b_last[t] : 0.f;
    Stream st;
    st.desc = i32x4{(int)(unsigned)(size_t)p.blob, (int)(((size_t)p.blob >> 32) & 0xFFFFu), (int)p.blob_bytes, 0x00020000};
    st.lds0 = (unsigned)(size_t)(LDS3 unsigned char*)ring;
    st.iss = !kSplitIssue || w < CIW;
    st.voff = (unsigned)(CPW * (st.iss ? w : 0)) * 1024u + (unsigned)lane * 16u;
    st.src0 = 0;
    st.n_slots = p.n_slots;
    const int my_passes = (p.n_pass - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    st.total = my_passes * p.n_slots;
    st.cur = 0;
    st.ridx = 0;
    st.pf_dst = 0; st.pf_src = 0; st.pf_voff = kOobC;
    // Z stages of this wave: [2][4 KB] behind the bias table
    const LDS3 unsigned char* stages = (const LDS3 unsigned char*)(lbias + 1056) + (size_t)w * 8192;
    const unsigned lds_stage = __builtin_amdgcn_readfirstlane((unsigned)(size_t)stages);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < CRING - 1; ++s) st.issue_whole(s);
    const i32x4 d_z = i32x4{(int)(unsigned)(size_t)q.Z, (int)(((size_t)q.Z >> 32) & 0xFFFFu), (int)(unsigned)((size_t)q.nq * q.ldz * 2), 0x00020000};

    // Z line L of the wave's 32 rows: instruction i fetches rows 8 i .. 8 i + 7 (8 lanes a row), lane position p of a row receiving source
    // chunk p ^ ((row >> 1) & 7): the ds_read_b128 lane groups then fall on 16 different bank quads
    auto z_offsets = [&](int pass, unsigned (&zo)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3);
            const long ql = (long)pass * 256 + w * 32 + r;
            zo[i] = (pass < p.n_pass && ql < q.nq) ? (unsigned)(ql * q.ldz * 2) + ((unsigned)((lane & 7) ^ ((r >> 1) & 7)) << 4) : kOobC;
        }
    };
    auto fetch_line = [&](const unsigned (&zo)[4], int L, int buf) {
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)L * 128u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_stage + (unsigned)buf * 4096u + (unsigned)i * 1024u);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" :: "v"(zo[i]), "s"(dst), "s"(d_z), "s"(so) : "memory");
        }
    };
    // this lane's B fragment of k-step sl (0..3) of a staged line: chunk 2 sl + lh of row li
    const LDS3 unsigned char* rd[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) rd[sl] = stages + li * 128 + ((((unsigned)(2 * sl + lh)) ^ (unsigned)((li >> 1) & 7)) << 4);

    unsigned zo[4];
    z_offsets(blockIdx.x, zo);
    fetch_line(zo, 0, 0);
    constexpr int NP = (PAIRS ? 3 : 1) * CPW;           // weight pieces a wave issues in a line step BEHIND its line fetch (which follows the first slot)

    int pass_i = 0;
#pragma unroll 1
    for (int pass = blockIdx.x; pass < p.n_pass; pass += gridDim.x, ++pass_i) {
        // ================= input layer: 8 accumulator tiles, Z streamed by lines ======================================================
        f32x16 acc8[8];
#pragma unroll
        for (int T = 0; T < 8; ++T) acc8[T] = bias_frag(lbias, T, lh);
#pragma unroll 1
        for (int L2 = 0; L2 < p.nline; L2 += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int L = L2 + par;
                // line L has landed: it was fetched behind the first slot of the previous line step (the first line of a pass: under the hidden
                // layers of the pass before), and this wave has issued NP weight pieces since
                if (L == 0 && pass_i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else vm_wait<NP, 0>(st.iss);
                u32x4 bz[4];
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) bz[sl] = *(const LDS3 u32x4*)(rd[sl] + par * 4096);
                // counted waits of the slots: behind slot k's pieces (issued in slot k - 3) this wave issued those of the next two slots and,
                // except for a line step's first slot (covered by the wait above), at least the 4 instructions of one line fetch
                if (!PAIRS) {
                    // (the wait above is the stricter one for a line step's first slot)
                    decode_in_slot<false, 0>(ring, st, lane, bz[0], bz[1], acc8);
                    fetch_line(zo, L + 1, par ^ 1);           // past the row's end: out of range, zeros (the counts stay the same)
                    decode_in_slot<false, 4>(ring, st, lane, bz[2], bz[3], acc8);
                } else {
                    decode_in_slot<true, 0>(ring, st, lane, bz[0], bz[0], acc8);
                    fetch_line(zo, L + 1, par ^ 1);
                    decode_in_slot<true, 4>(ring, st, lane, bz[1], bz[1], acc8);
                    decode_in_slot<true, 4>(ring, st, lane, bz[2], bz[2], acc8);
                    decode_in_slot<true, 4>(ring, st, lane, bz[3], bz[3], acc8);
                }
            }
        }
        // ================= hidden layers ================================================================================================
        u32x4 act0[CM][16], act1[CM][16];
        f32x16 acc[2][CM];
#pragma unroll
        for (int T = 0; T < 7; ++T)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) finish_quad(T, acc8[T], act0[0], qd);
        acc[1][0] = acc8[7];                                    // (the first hidden tile finishes tile 7, as in the kv chains)
        unsigned zn[4];
        z_offsets(pass + (int)gridDim.x, zn);
        fetch_line(zn, 0, 0);                                   // the next pass's first line: 4 operations in front of the hidden chain (BX)
        LogitAcc lg;
        lg.on = false;
        Layers<PAIRS>::template layer<0, false, 4>(ring, st, act0, act1, lbias + 256, acc, lane, &lg);
        Layers<PAIRS>::template layer<1>(ring, st, act0, act1, lbias + 256, acc, lane, &lg);
        Layers<PAIRS>::template layer<2>(ring, st, act0, act1, lbias + 256, acc, lane, &lg);
        finish_tile(7, acc[1], act1);
        // ================= output layer: one tile, hi + lo halves of W5 ================================================================
        const f32x16 c0 = bias_frag(lbias + 1024, 0, lh);
        tile_mma<true, -1>(ring, lane, act1, c0, acc[0], acc[1], act0, st);
        if (lh == 0) {
            const long ql = (long)pass * 256 + w * 32 + li;
            if (ql < q.nq) {
                const long qq = q.q0 + ql;
                float v[3] = {acc[0][0][0], acc[0][0][1], acc[0][0][2]};
                if (q.x_lr) {
                    const float cy = q.coord[2 * qq], cx = q.coord[2 * qq + 1];
                    float fy = sub_rn(mul_rn(add_rn(cy, 1.0f), (float)q.H * 0.5f), 0.5f);
                    float fx = sub_rn(mul_rn(add_rn(cx, 1.0f), (float)q.W * 0.5f), 0.5f);
                    fy = fminf((float)(q.H - 1), fmaxf(fy, 0.f));
                    fx = fminf((float)(q.W - 1), fmaxf(fx, 0.f));
                    const float y0f = floorf(fy), x0f = floorf(fx);
                    const int y0 = (int)y0f, x0 = (int)x0f;
                    const float wy1 = fy - y0f, wy0 = (y0f + 1.f) - fy;
                    const float wx1 = fx - x0f, wx0 = (x0f + 1.f) - fx;
                    const int y1 = min(y0 + 1, q.H - 1), x1 = min(x0 + 1, q.W - 1);     // weights of clamped taps are 0
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float* img = q.x_lr + (size_t)c * q.H * q.W;
                        v[c] += img[(size_t)y0 * q.W + x0] * (wx0 * wy0) + img[(size_t)y0 * q.W + x1] * (wx1 * wy0) +
                                img[(size_t)y1 * q.W + x0] * (wx0 * wy1) + img[(size_t)y1 * q.W + x1] * (wx1 * wy1);
                    }
                }
                q.rgb[qq * 3] = v[0];
                q.rgb[qq * 3 + 1] = v[1];
                q.rgb[qq * 3 + 2] = v[2];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) zo[i] = zn[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA of this workgroup may land after it has gone
}

}  // namespace chain

// ---- host side ------------------------------------------------------------------------------------------------------------------------
static int chain_tiles(const ciaosr_head_weights_t* w, int pairs) {
    const int n_vout = (w->v.width[w->v.n_layers - 1] + 31) / 32;
    (void)pairs;
    return 24 + 24 + ((n_vout + 1) & ~1);           // the v-out loop runs two units per iteration: an odd count is padded with a zero tile
}

bool head_chain_ok(const ciaosr_head_weights_t* w) {
    if (w->k.n_layers != 5 || w->v.n_layers != 5 || w->local_size != 2) return false;
    for (int i = 0; i < 4; ++i)
        if (w->k.width[i] != 256 || w->v.width[i] != 256) return false;
    const int n_vout = (w->v.width[4] + 31) / 32;
    const size_t lds = (size_t)chain::CRING * chain::CSLOT + 2 * chain::CTAIL + (size_t)(1536 + 32 * (n_vout + 1)) * 4 + (size_t)chain::CNW * chain::CM * chain::CNSTAGE * chain::CSTAGE;
    return lds <= 160 * 1024;
}

bool head_decode_chain_ok(const ciaosr_head_weights_t* w);
size_t head_decode_chain_bytes(const ciaosr_head_weights_t* w, int pairs);
int pack_head_decode_chain(const ciaosr_head_weights_t* w, int pairs, void* out, hipStream_t s);
// bytes of the kv part of the blob (the imnet_q stream, if its shape allows one, follows it)
size_t head_kv_chain_bytes(const ciaosr_head_weights_t* w, int pairs) {
    return 2 * (size_t)chain::CTAIL + (size_t)chain_tiles(w, pairs) * chain::CTILE * (pairs ? 2 : 1);
}
size_t head_chain_bytes(const ciaosr_head_weights_t* w, int pairs) {
    return head_kv_chain_bytes(w, pairs) + head_decode_chain_bytes(w, pairs);
}

int pack_head_chain(const ciaosr_head_weights_t* w, int pairs, void* out, hipStream_t s) {
    using namespace chain;
    unsigned char* o = reinterpret_cast<unsigned char*>(out);
    const int D = (w->no_unfold ? 1 : 9) * w->channels, Dv = D + w->nonlocal_channels;
    hipLaunchKernelGGL(pack_tail_kernel, dim3(2), dim3(256), 0, s, w->k.weight[0], w->k.ld[0], D, reinterpret_cast<uint4*>(o));
    hipLaunchKernelGGL(pack_tail_kernel, dim3(2), dim3(256), 0, s, w->v.weight[0], w->v.ld[0], Dv, reinterpret_cast<uint4*>(o + CTAIL));
    const size_t tile_bytes = (size_t)CTILE * (pairs ? 2 : 1);
    unsigned char* st = o + 2 * CTAIL;
    const int total = chain_tiles(w, pairs);
    if (hipMemsetAsync(st, 0, (size_t)total * tile_bytes, s) != hipSuccess) return CIAOSR_ERR_LAUNCH;
    int tile = 0;
    for (int c = 0; c < 2; ++c) {
        const ciaosr_mlp_t& m = c == 0 ? w->k : w->v;
        for (int l = 1; l <= 3; ++l) {
            hipLaunchKernelGGL(pack_chain_kernel, dim3(64), dim3(256), 0, s, m.weight[l], m.ld[l], 256, 8, pairs, reinterpret_cast<uint4*>(st + (size_t)tile * tile_bytes));
            tile += 8;
        }
    }
    const int n_vout = (w->v.width[4] + 31) / 32;
    hipLaunchKernelGGL(pack_chain_kernel, dim3(128), dim3(256), 0, s, w->v.weight[4], w->v.ld[4], w->v.width[4], n_vout, pairs,
                       reinterpret_cast<uint4*>(st + (size_t)tile * tile_bytes));
    const int rc = launch_status("pack_head_chain" CIAOSR_H16_SUFFIX);
    if (rc != CIAOSR_OK) return rc;
    return pack_head_decode_chain(w, pairs, o + head_kv_chain_bytes(w, pairs), s);
}

// kp: the FusedKVP of the 128-row kernel (tables, G, Z, biases); blob: pack_head_chain's output
int head_kv_chain_h16(const FusedKVP& kp, const ciaosr_head_weights_t* w, const void* blob, int pairs, int grid_w, int* flag, hipStream_t s) {
    using namespace chain;
    ChainP p;
    p.kv = kp;
    p.blob = reinterpret_cast<const unsigned char*>(blob);
    p.blob_bytes = (unsigned)head_kv_chain_bytes(w, pairs);
    p.n_slots = chain_tiles(w, pairs) * (pairs ? 2 : 1);
    p.n_vout = (w->v.width[4] + 31) / 32;
    p.grid_w = (grid_w > 0 && kp.nq % grid_w == 0 && kp.q0 % grid_w == 0) ? grid_w : 0;
    if (p.grid_w) p.n_pass = ((p.grid_w + 15) / 16) * ((kp.nq / p.grid_w + 3) / 4);
    else p.n_pass = (kp.nq + CQ - 1) / CQ;
    p.flag = flag;
    const size_t lds = (size_t)CRING * CSLOT + 2 * CTAIL + (size_t)(1536 + 32 * (p.n_vout + 1)) * 4 + (size_t)CNW * CM * CNSTAGE * CSTAGE;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = p.n_pass < cus ? p.n_pass : cus;
    ProfScope prof(pairs ? "head_kv_chain_pairs" CIAOSR_H16_SUFFIX : "head_kv_chain" CIAOSR_H16_SUFFIX, s);
    if (pairs) {
        CIAOSR_BIG_LDS(head_kv_chain_kernel<true>, lds);
        hipLaunchKernelGGL(head_kv_chain_kernel<true>, dim3(grid), dim3(64 * CNW), lds, s, p);
    } else {
        CIAOSR_BIG_LDS(head_kv_chain_kernel<false>, lds);
        hipLaunchKernelGGL(head_kv_chain_kernel<false>, dim3(grid), dim3(64 * CNW), lds, s, p);
    }
    return launch_status("head_kv_chain" CIAOSR_H16_SUFFIX);
}

// ---- imnet_q stream ---------------------------------------------------------------------------------------------------------------------
bool head_decode_chain_ok(const ciaosr_head_weights_t* w) {
    const ciaosr_mlp_t& m = w->q;
    if (m.n_layers != 5 || m.width[4] != 3) return false;
    for (int i = 0; i < 4; ++i)
        if (m.width[i] != 256) return false;
    const int Dv = (w->no_unfold ? 1 : 9) * w->channels + w->nonlocal_channels;
    return Dv > 0 && Dv % 128 == 0;                 // whole 128-B Z lines, an even number of them
}
static int decode_slots(const ciaosr_head_weights_t* w, int pairs) {
    const int Dv = (w->no_unfold ? 1 : 9) * w->channels + w->nonlocal_channels;
    const int nks = Dv / 16;
    return pairs ? nks + 48 + 2 : nks / 2 + 24 + 2;
}
size_t head_decode_chain_bytes(const ciaosr_head_weights_t* w, int pairs) {
    return head_decode_chain_ok(w) ? (size_t)decode_slots(w, pairs) * chain::CSLOT : 0;
}
int pack_head_decode_chain(const ciaosr_head_weights_t* w, int pairs, void* out, hipStream_t s) {
    using namespace chain;
    if (!head_decode_chain_ok(w)) return CIAOSR_OK;
    const ciaosr_mlp_t& m = w->q;
    unsigned char* o = reinterpret_cast<unsigned char*>(out);
    const int Dv = (w->no_unfold ? 1 : 9) * w->channels + w->nonlocal_channels;
    const int nks = Dv / 16;
    hipLaunchKernelGGL(pack_decode_in_kernel, dim3(256), dim3(256), 0, s, m.weight[0], m.ld[0], nks, pairs, reinterpret_cast<uint4*>(o));
    o += (size_t)(pairs ? nks : nks / 2) * CSLOT;
    const size_t tile_bytes = (size_t)CTILE * (pairs ? 2 : 1);
    for (int l = 1; l <= 3; ++l) {
        hipLaunchKernelGGL(pack_chain_kernel, dim3(64), dim3(256), 0, s, m.weight[l], m.ld[l], 256, 8, pairs, reinterpret_cast<uint4*>(o));
        o += 8 * tile_bytes;
    }
    hipLaunchKernelGGL(pack_chain_kernel, dim3(8), dim3(256), 0, s, m.weight[4], m.ld[4], 3, 1, 1, reinterpret_cast<uint4*>(o));      // hi + lo always
    return launch_status("pack_head_decode_chain" CIAOSR_H16_SUFFIX);
}

// qp: the FusedQP of the 128-row kernel; blob: pack_head_decode_chain's output
int head_decode_chain_h16(const FusedQP& qp, const ciaosr_head_weights_t* w, const void* blob, int pairs, hipStream_t s) {
    using namespace chain;
    DecodeChainP p;
    p.q = qp;
    p.blob = reinterpret_cast<const unsigned char*>(blob);
    p.blob_bytes = (unsigned)head_decode_chain_bytes(w, pairs);
    p.n_slots = decode_slots(w, pairs);
    p.nline = qp.Dv / 64;
    p.n_pass = (qp.nq + 255) / 256;
    const size_t lds = (size_t)CRING * CSLOT + 1056 * 4 + (size_t)CNW * 8192;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = p.n_pass < cus ? p.n_pass : cus;
    ProfScope prof(pairs ? "head_decode_chain_pairs" CIAOSR_H16_SUFFIX : "head_decode_chain" CIAOSR_H16_SUFFIX, s);
    if (pairs) {
        CIAOSR_BIG_LDS(head_decode_chain_kernel<true>, lds);
        hipLaunchKernelGGL(head_decode_chain_kernel<true>, dim3(grid), dim3(64 * CNW), lds, s, p);
    } else {
        CIAOSR_BIG_LDS(head_decode_chain_kernel<false>, lds);
        hipLaunchKernelGGL(head_decode_chain_kernel<false>, dim3(grid), dim3(64 * CNW), lds, s, p);
    }
    return launch_status("head_decode_chain" CIAOSR_H16_SUFFIX);
}

}  // namespace CIAOSR_H16_NS
}  // namespace ciaosr

#if defined(CIAOSR_PROBE) && CIAOSR_F16
extern "C" int ciaosr_debug_probe_chain_read(unsigned long long* host, int n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ciaosr::f16::chain::g_cprobe), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
#endif
