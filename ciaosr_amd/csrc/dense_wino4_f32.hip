// fp32 dense-block convolution for big maps in Winograd F(4x4, 3x3) form (round 4): the same 3x3 dense layer as dense_wino_f32.hip
// (mmedit RDB.layers[l].conv over cat(x, d_0 .. d_{l-1}), called from ciaosr_net.py:330-337) with 36 multiplies per 4x4 output tile and
// (ci, co) pair instead of 144 (direct) or 64 (F(2x2, 3x3)): 2.25 MFMA-units per output pixel against 4.
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A        g 3x3 weights, d the 6x6 input tile whose top-left is the output tile's (-1, -1)
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]      (applied on the host in fp64, rounded once)
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// fp32 throughout on the exact-fp32 MFMA.  Not bitwise a direct fmaf chain (products of transformed operands; the transforms multiply
// by up to 5 / 8): measured against the oracle in the tests, selectable against F(2x2) and the direct kernel per call.
//
// Why it is cut differently from the F(2x2) kernel.  There a wave transforms its own operands in registers beside its MFMAs; with 36
// positions the six-row window transient does not fit beside the accumulators, and a VALU instruction beside an fp32
// MFMA costs the SIMD's pipe ~6-20 cycles (tools/ubench/mfma_valu.hip).  So the two kinds of work are SEPARATED IN TIME:
//   * workgroup = 16 x 32 output pixels = 4 x 8 Winograd tiles = the 32 columns of one MFMA tile, x all 64 output channels; 4 waves, one
//     per SIMD; a step = 8 input channels;
//   * T phase (all 256 threads, thread = (tile, channel)): B^T d B of the thread's 6x6 window -- the 12-operation form of B^T x on 6
//     columns (as column PAIRS in packed fp32) and 6 rows: 108 VALU instructions -- and 36 dword LDS writes into V[pos][tile][8 ch].
//     No MFMA is in flight (the VALU instructions cost their own 4 cycles) and NO memory request is issued: the window values were
//     read from the patch during the previous M phase (the same reads at the top of this phase cost 800 cycles of a 6800-cycle step);
//   * one barrier;
//   * M phase: wave (nt, h) owns positions (3h .. 3h+2, 0 .. 5) of output-channel half nt = 18 accumulator tiles (288 registers): 18
//     V fragments (ds_read_b128) against 18 weight fragments (1 KB each, straight from L2) = 72 MFMAs.  One wave per SIMD issues in
//     order: a memory instruction placed behind a BLOCK of MFMAs is issued once the block has been issued, and the next MFMA waits for
//     it (first version: 1570 cycles for the 18 weight requests, 970 for the 5 DMA pieces of an 8000-cycle step).  So every memory
//     instruction of a step sits behind ONE MFMA (w4_slot_plan below): V fragments of the next row, window values of the next step,
//     row 2's weights of THIS step (inside row 0: two rows to land), the DMA pieces of patch n + 2, rows 0 / 1's weights of step n + 1,
//     each weight register re-requested >= 8 MFMAs behind its last use;
//   * the 18 x 34-pixel halo patch of a step's 8 channels (2 x 16-B chunks per pixel) comes by LDS-DMA two steps ahead into one of two
//     buffers; its chunks are XOR-swizzled with ((P >> 2) & 3) << 1 (P = pixel index) so that the window reads -- four tiles x eight
//     channels per 32-lane group, tiles four pixels apart -- hit 32 distinct banks;
//   * V and the patch are double-buffered, which leaves ONE barrier per step (T(n) | barrier | M(n), T(n+1) | barrier | ...).
// Measured (tools/wino4_probe.py, tools/wino4_fit.py): 6103 cycles per step = T 580 + barrier 77 + M 5505 (MFMA issue 4608).
// The output transform runs once per layer: A^T along j in registers, the row half through LDS between the two waves of a channel half.
#include "ops.h"
#include <type_traits>

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int W4TH = 16, W4TW = 32;              // output tile (pixels)
constexpr int W4PH = W4TH + 2, W4PW = W4TW + 2;  // patch with the 1-pixel halo
constexpr int W4NP = W4PH * W4PW;                // 612 pixels
constexpr int W4PIECES = (2 * W4NP + 63) / 64;   // 20 DMA pieces of 64 chunks (16 B each)
constexpr int W4PATCH = W4PIECES * 1024;         // 20 480 B per buffer
constexpr int W4VROW = 32;                       // bytes per (position, tile) row of V: 8 channels = two 16-B chunks; chunk (2 tile + half) is kept at
                                                 // chunk ^ ((tile >> 3) & 1): the M phase's b128 reads (16 lanes = 16 tiles of one half per pass)
                                                 // and the T phase's dword writes (8 tiles x 8 channels per wave) both touch every bank once
constexpr int W4VPOS = 32 * W4VROW;              // 1024
constexpr int W4V = 36 * W4VPOS;                 // 36 864 B per buffer
constexpr int W4V0 = 2 * W4PATCH;                // LDS: [patch 0][patch 1][V 0][V 1]
constexpr size_t kWino4Xch = 4 * 8 * 4 * 64 * 16;                        // output-transform exchange: 131 072 B
constexpr size_t kWino4Loop = 2 * (size_t)W4PATCH + 2 * (size_t)W4V;     // 114 688 B
constexpr size_t kWino4Lds = kWino4Loop > kWino4Xch ? kWino4Loop : kWino4Xch;
constexpr unsigned kOobW4 = 0xFFFFFFF0u;

struct DenseWino4P {
    float* x; int ldx;                           // the block buffer [n_img][H*W][ldx]: input groups and the output group
    unsigned x_bytes;
    int H, W, tiles_x;
    int groups;                                  // 64-channel input groups
    const float4* wf;                            // 36 fragment arrays [2][nj][64 lanes] float4, one per transformed position, back to back
    int nj; long pos_stride;                     // nj = cin / 8; float4 per position array
    const float* bias;                           // nullptr (TABLE): no bias, no ReLU
    int col_out;
    // TABLE form (the fused head's logit table, nine 3x3 convolutions of product maps without bias): the output goes elsewhere
    float* out; unsigned out_bytes;              // [H*W][ld_out] per (image = map o, channel block z): byte offset o * out_img_bytes, channel 64 z
    int ld_out; unsigned out_img_bytes;
    unsigned wf_block_bytes;                     // weight fragments of channel block z start wf_block_bytes * z bytes further
};

// What goes out behind the MFMA of slot `slot` (0 .. 23) of row r3 (0 .. 2) of an M phase.  Per step and wave: 6 + 6 V fragments (rows 1, 2;
// row 0's are read in front of the row), 36 window values of the next step, 6 weight requests of row 2 of THIS step, 5 DMA pieces, 12
// weight requests of rows 0 / 1 of the next step = 71 memory instructions behind 72 MFMAs.  Order constraints: row 2's weights go out
// BEFORE the DMA pieces (the wait in front of row 2 then does not ask for the patch), a row's weights are re-requested >= 8 MFMAs behind
// their last use, the last window value is read >= 8 MFMAs before the T phase.
struct W4Slot { int vfrag, window, w2, dma, wnext; };
constexpr W4Slot w4_slot_plan(int r3, int slot) {
    W4Slot s = {-1, -1, -1, -1, -1};
    if (r3 < 2 && slot % 4 == 0) s.vfrag = slot / 4;                          // rows 0 and 1: 0 4 .. 20
    if (r3 < 2 && slot % 2 == 1) s.window = 12 * r3 + slot / 2;               // rows 0 and 1: the 12 odd slots -> 0 .. 23
    if (r3 == 0 && slot < 6) s.w2 = slot;                                     // 0 .. 5
    if (r3 == 0 && slot >= 6 && slot % 4 == 2) s.dma = (slot - 6) / 4;        // 6 10 14 18 22
    if (r3 == 1 && slot % 4 == 2) s.wnext = slot / 4;                         // row 0's weights: 2 6 .. 22
    if (r3 == 2 && slot < 18) {
        if (slot % 3 == 2) s.wnext = 6 + slot / 3;                            // row 1's weights: 2 5 .. 17
        else s.window = 24 + 2 * (slot / 3) + slot % 3;                       // 12 slots -> 24 .. 35
    }
    return s;
}
// (A plan with at most one memory instruction per slot -- row 2's weights on row 0's even slots, the window reads spread over all
// three rows -- measured 90 cycles per step SLOWER; what an M phase pays for is the vector-memory instruction itself, ~70 cycles each
// even behind an MFMA: the same six requests cost 340 cycles at the top of the T phase and 440 inside row 0.)

// y = B^T x for the 6-vector x (12 operations)
#define W4_BT(x0, x1, x2, x3, x4, x5, y0, y1, y2, y3, y4, y5)                          \
    do {                                                                               \
        const float t1_ = __builtin_fmaf(-4.f, x2, x4), t2_ = __builtin_fmaf(-4.f, x1, x3); \
        const float t3_ = x4 - x2, t4_ = x3 - x1;                                      \
        y0 = __builtin_fmaf(4.f, x0, __builtin_fmaf(-5.f, x2, x4));                   \
        y1 = t1_ + t2_;                                                                \
        y2 = t1_ - t2_;                                                                \
        y3 = __builtin_fmaf(2.f, t4_, t3_);                                            \
        y4 = __builtin_fmaf(-2.f, t4_, t3_);                                           \
        y5 = __builtin_fmaf(4.f, x1, __builtin_fmaf(-5.f, x3, x5));                   \
    } while (0)

#ifdef CIAOSR_PROBE      // developer probe build (make probe; tools/wino4_probe.py): s_memtime stamps of the last launch's workgroups
__device__ unsigned long long g_w4probe[1024 * 8];
#define W4PROBE(slot) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 1024) g_w4probe[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#define W4PROBE_ADD(slot, t0) do { w4acc##slot += __builtin_readcyclecounter() - (t0); } while (0)
#else
#define W4PROBE(slot) do { } while (0)
#define W4PROBE_ADD(slot, t0) do { } while (0)
#endif

template <bool TABLE>
__global__ __launch_bounds__(256) void dense_wino4_f32_kernel(DenseWino4P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds4[];
    W4PROBE(0);
#ifdef CIAOSR_PROBE
    unsigned long long w4acc5 = 0, w4acc6 = 0;
#endif
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, lh = lane >> 5;
    const int nt = w & 1, h = w >> 1;            // M phase: output-channel half, rows 3h .. 3h+2 of the 6x6 transformed domain
    const int ty0 = (blockIdx.x / p.tiles_x) * W4TH, tx0 = (blockIdx.x % p.tiles_x) * W4TW;
    const int img = blockIdx.y;
    const unsigned img_off = (unsigned)((size_t)img * p.H * p.W * p.ldx * 4);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds4;

    // ---- patch DMA: piece i = w + 4 s (i < 20) fills LDS bytes [1024 i, 1024 i + 1024) of a patch buffer; lane -> LDS chunk q' = 64 i + lane,
    // which holds chunk q = q' ^ (((q' >> 3) & 3) << 1) of the unswizzled order q = 2 P + half (pixel P = py * 34 + px, half = channels 4 half ..)
    constexpr int W4DS = W4PIECES / 4;           // 5 per wave
    unsigned goff[W4DS];
#pragma unroll
    for (int s = 0; s < W4DS; ++s) {
        const int qs = 64 * (w + 4 * s) + lane;
        const int q = qs ^ (((qs >> 3) & 3) << 1);
        const int P = q >> 1, half = q & 1;
        const int py = P / W4PW, px = P - py * W4PW;
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
        const bool ok = P < W4NP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        goff[s] = ok ? (img_off + (unsigned)(gy * p.W + gx) * (unsigned)p.ldx * 4u + (unsigned)half * 16u) : kOobW4;
    }
    const int rot = p.groups > 1 ? (int)(blockIdx.x % (unsigned)p.groups) : 0;
    auto phys = [&](int g) -> int { const int x = g + rot; return x >= p.groups ? x - p.groups : x; };
    const i32x4 desc = {(int)(unsigned)(size_t)p.x, (int)(((size_t)p.x >> 32) & 0xFFFFu), (int)p.x_bytes, 0x00020000};
    const int N = 8 * p.groups;                  // steps
    auto step_ch = [&](int n) -> int { return 64 * phys(n >> 3) + 8 * (n & 7); };     // first input channel of step n
    auto dma_patch = [&](int n) {                // patch of step n -> buffer n & 1
        const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)step_ch(n) * 4u);
#pragma unroll
        for (int s = 0; s < W4DS; ++s) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((n & 1) * W4PATCH) + 1024u * (unsigned)(w + 4 * s));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(goff[s]), "s"(dst), "s"(desc), "s"(soff) : "memory");
        }
    };

    // ---- T phase addressing: thread = (tile, channel); window pixel (a, b) of the tile at P = P0 + 34 a + b, P0 = 136 tyw + 4 txw (a
    // multiple of 4), chunk q = 2 P + half, swizzle key ((P >> 2) & 3) << 1 = ((k0 + ((34 a + b) >> 2)) & 3) << 1 with k0 = (P0 >> 2) & 3:
    // byte address = tbase + 32 ((34 a + b) & ~3) + 32 (((34 a + b) & 3) ^ K_c), K_c = (k0 + c) & 3, c = ((34 a + b) >> 2) & 3.
    // The 16 values ty[m][c] = tbase + 32 (m ^ K_c) are kept in registers; the rest is an immediate.
    const int tch = t & 7, ttile = t >> 3;
    const int ttyw = ttile >> 3, ttxw = ttile & 7;
    int ty[4][4];
    {
        const int P0 = 4 * W4PW * ttyw + 4 * ttxw;
        const int k0 = (P0 >> 2) & 3;
        const int tbase = 32 * P0 + 16 * (tch >> 2) + 4 * (tch & 3);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int c = 0; c < 4; ++c) ty[m][c] = tbase + 32 * (m ^ ((k0 + c) & 3));
    }
    const int tvoff = ((2 * ttile + (tch >> 2)) ^ ((ttile >> 3) & 1)) * 16 + (tch & 3) * 4;     // + pos * 1024: this thread's element of V

    // ---- weights: fragment (position 18 h + pp, nt) of k-chunk jc, through a buffer descriptor with the fragment index as scalar offset.
    // The loads are INLINE ASM and their waits are counted by hand: hipcc cannot see the patch DMA (asm as well) in the vector-memory
    // queue, and where its own count of the weight loads merges over the loop's paths it falls back to `s_waitcnt vmcnt(0)` in front
    // of an MFMA -- which, vmcnt retiring in order, waits for the DMA pieces issued a few MFMAs earlier to come back from HBM.
    // Queue order in the steady state (oldest first) at the top of M(n): W0(n) W1(n) [6 loads each: rows 0 / 1 of step n]; row 0
    // requests W2(n) (row 2 of the SAME step: its registers are free since the end of M(n - 1), and two rows = ~3000 cycles cover an L2
    // round trip) and then the 5 DMA pieces of patch n + 2, row 1 re-requests W0(n + 1), row 2 W1(n + 1).  The T phase carries no
    // vector-memory request: at its top six of them cost ~340 cycles of issue (one wave per SIMD, nothing to hide them behind), behind
    // its V writes they filled the 4 waves' shared address unit right in front of the M phase (+450 cycles there).
    const i32x4 wdesc = {(int)(unsigned)(size_t)p.wf, (int)(((size_t)p.wf >> 32) & 0xFFFFu), (int)0xFFFFFFFFu, 0x00020000};
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned pos_bytes = (unsigned)p.pos_stride * 16u;
    auto wload = [&](f32x4& dst, int pp, int n) __attribute__((always_inline)) {
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(18 * h + pp) * pos_bytes + (unsigned)(nt * p.nj + (step_ch(n) >> 3)) * 1024u +
                                                           (TABLE ? (unsigned)blockIdx.z * p.wf_block_bytes : 0u));
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(lane16), "s"(wdesc), "s"(so) : "memory");
    };

    // prologue: patches of steps 0 and 1, weights of step 0; the accumulators are cleared under their latency
    dma_patch(0);
    if (N > 1) dma_patch(1);
    f32x4 wr[18];
#pragma unroll
    for (int pp = 0; pp < 12; ++pp) wload(wr[pp], pp, 0);       // rows 0 and 1; row 2's are requested inside row 0 of every step
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc[18];
#pragma unroll
    for (int pp = 0; pp < 18; ++pp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pp][e] = 0.f;
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");           // the DMA pieces are older than the 12 weight requests
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    W4PROBE(1);
    // The 36 window values of this thread's (tile, channel) of step n + 1 are read from the patch DURING M(n), one ds_read behind an
    // MFMA each (measured: the same reads at the top of the T phase cost 800 of a 6800-cycle step -- LDS latency and the 4 waves'
    // 2-way bank conflicts with nothing to cover them).  Patch n + 1 has landed by the barrier in front of M(n).
    float dn[36];
    auto patch_value = [&](int n, int o36) __attribute__((always_inline)) -> float {
        const int a = o36 / 6, b = o36 - 6 * a, o = W4PW * a + b;
        return *reinterpret_cast<const float*>(lds4 + (n & 1) * W4PATCH + ty[o & 3][(o >> 2) & 3] + 32 * (o & ~3));
    };
#pragma unroll
    for (int o36 = 0; o36 < 36; ++o36) dn[o36] = patch_value(0, o36);
    const int vfo = ((2 * li + lh) ^ ((li >> 3) & 1)) * 16;      // this lane's float4 of a V row (M phase)
#pragma unroll 1
    for (int n = 0; n < N; ++n) {
        // ================= T(n): B^T d B of this thread's (tile, channel) =========================================================
#ifdef CIAOSR_PROBE
        const unsigned long long tstep = __builtin_readcyclecounter();
#endif
        {
            unsigned char* vb = lds4 + W4V0 + (n & 1) * W4V + tvoff;
            float d[6][6];                                       // read during M(n - 1) (step 0: behind the prologue's barrier)
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) d[a][b] = dn[6 * a + b];
            // pass 1 (down the columns) on column PAIRS in packed fp32: 36 instructions instead of 72 (same lanes of both halves: no
            // crossed source selection, Makefile).  Pass 2 runs along the pairs and stays scalar.
            float r[6][6];
#pragma unroll
            for (int bp = 0; bp < 3; ++bp) {
                f32x2 x[6], y[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) x[a] = f32x2{d[a][2 * bp], d[a][2 * bp + 1]};
                const f32x2 t1 = __builtin_elementwise_fma(f32x2{-4.f, -4.f}, x[2], x[4]), t2 = __builtin_elementwise_fma(f32x2{-4.f, -4.f}, x[1], x[3]);
                const f32x2 t3 = x[4] - x[2], t4 = x[3] - x[1];
                y[0] = __builtin_elementwise_fma(f32x2{4.f, 4.f}, x[0], __builtin_elementwise_fma(f32x2{-5.f, -5.f}, x[2], x[4]));
                y[1] = t1 + t2;
                y[2] = t1 - t2;
                y[3] = __builtin_elementwise_fma(f32x2{2.f, 2.f}, t4, t3);
                y[4] = __builtin_elementwise_fma(f32x2{-2.f, -2.f}, t4, t3);
                y[5] = __builtin_elementwise_fma(f32x2{4.f, 4.f}, x[1], __builtin_elementwise_fma(f32x2{-5.f, -5.f}, x[3], x[5]));
#pragma unroll
                for (int i = 0; i < 6; ++i) { r[i][2 * bp] = y[i][0]; r[i][2 * bp + 1] = y[i][1]; }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float v0, v1, v2, v3, v4, v5;
                W4_BT(r[i][0], r[i][1], r[i][2], r[i][3], r[i][4], r[i][5], v0, v1, v2, v3, v4, v5);
                *reinterpret_cast<float*>(vb + (6 * i + 0) * W4VPOS) = v0;
                *reinterpret_cast<float*>(vb + (6 * i + 1) * W4VPOS) = v1;
                *reinterpret_cast<float*>(vb + (6 * i + 2) * W4VPOS) = v2;
                *reinterpret_cast<float*>(vb + (6 * i + 3) * W4VPOS) = v3;
                *reinterpret_cast<float*>(vb + (6 * i + 4) * W4VPOS) = v4;
                *reinterpret_cast<float*>(vb + (6 * i + 5) * W4VPOS) = v5;
            }
        }
        // every DMA piece of patch n + 1 (issued before the 12 weight requests of rows 1 and 2 of the previous M phase) has landed
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        W4PROBE_ADD(5, tstep);                   // T phase up to the barrier
        __syncthreads();
        W4PROBE_ADD(6, tstep);                   // ... and through it
        __builtin_amdgcn_sched_barrier(0);
        // ================= M(n): 18 positions x 4 MFMAs of this wave =================================================================
        // One wave per SIMD issues in order: a memory instruction placed BEHIND a block of MFMAs is issued only once the block has
        // been issued, and the next MFMA waits for it -- measured (ablations): the 18 weight reloads of a step cost 1570 cycles, the
        // 5 DMA pieces 970, of an 8000-cycle step with 4608 cycles of MFMA issue.  So every memory instruction of the step is placed
        // right behind ONE MFMA (whose 64 pipe cycles cover its issue): the V fragments of row r + 1 and the DMA pieces of patch
        // n + 2 (buffer n & 1: free, every wave is past T(n)) inside row 0, the reloads of row r - 1's weights inside row r; row 2's
        // weights are re-requested at the top of the next T phase, under its LDS reads.  VMEM order: DMA pieces, then 18 reloads.
        {
            const unsigned char* vb = lds4 + W4V0 + (n & 1) * W4V + 18 * h * W4VPOS + vfo;
            float4 vf[2][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) vf[0][j] = *reinterpret_cast<const float4*>(vb + j * W4VPOS);
            // Past the last step the requests repeat step N - 1 (same weights, the same patch into a buffer nobody reads any more): every
            // step then issues the same 23 vector-memory requests and the waits below are constants.
            const int n1 = n + 1 < N ? n + 1 : N - 1, n2 = n + 2 < N ? n + 2 : N - 1;
            const unsigned dsoff = __builtin_amdgcn_readfirstlane((unsigned)step_ch(n2) * 4u);
            const unsigned ddst = lds0 + (unsigned)((n & 1) * W4PATCH) + 1024u * (unsigned)w;
            auto mma = [&](int pp, float a, float bq) __attribute__((always_inline)) {
                // 18 accumulator tiles are 288 registers, 32 more than the accumulator file: tiles 16 and 17 are PINNED in VGPRs (asm
                // operand class "v"), the other sixteen fill the 256 AGPRs (left to itself hipcc shuffles accumulators between the files)
                if (pp < 16) acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc[pp], 0, 0, 0);
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[pp]) : "v"(a), "v"(bq));
            };
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) {
                // this row's weights have landed once at most the younger requests are outstanding: row 0: W1 = 6; row 1: row 0's W2 + 5
                // DMA pieces = 11; row 2: the DMA pieces + W0(n + 1) = 11 (W2 goes out in slots 0 .. 5, AHEAD of the DMA pieces, so that
                // this wait does not ask for the patch to be back from HBM).
                if (r3 == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int slot = 0; slot < 24; ++slot) {
                    const int c = slot / 6, j = slot % 6, pp = 6 * r3 + j;
                    const f32x4 wv = wr[pp];
                    const float4 bv = vf[r3 & 1][j];
                    mma(pp, c == 0 ? wv.x : c == 1 ? wv.y : c == 2 ? wv.z : wv.w, c == 0 ? bv.x : c == 1 ? bv.y : c == 2 ? bv.z : bv.w);
                    // ---- the memory instruction(s) of this slot (w4_slot_plan): at most one vector-memory request per slot, and an LDS read
                    // shares a slot with one only where the table says so
                    const W4Slot sp = w4_slot_plan(r3, slot);        // folded: both loops are fully unrolled
                    if (sp.vfrag >= 0)                            // V fragment of the NEXT row
                        vf[(r3 + 1) & 1][sp.vfrag] = *reinterpret_cast<const float4*>(vb + (6 * (r3 + 1) + sp.vfrag) * W4VPOS);
                    if (sp.window >= 0) dn[sp.window] = patch_value(n + 1, sp.window);    // window value of step n + 1 (past the end: stale, unused)
                    if (sp.w2 >= 0) wload(wr[12 + sp.w2], 12 + sp.w2, n);               // row 2's weights of THIS step: two rows to land
                    if (sp.dma >= 0) {                                                    // DMA piece w + 4 k of patch n + 2
                        const unsigned dst = __builtin_amdgcn_readfirstlane(ddst + 4096u * (unsigned)sp.dma);
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(goff[sp.dma]), "s"(dst), "s"(desc), "s"(dsoff) : "memory");
                    }
                    if (sp.wnext >= 0) wload(wr[sp.wnext], sp.wnext, n1);                // weights of step n + 1, >= 8 MFMAs behind their last use
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // The repeated requests of the last step must have landed INSIDE the loop.  Behind it the 18 weight registers are dead for
            // hipcc, which hands them to the output transform's temporaries -- and register-only code moves freely across an asm wait
            // placed behind the loop, so the late loads landed in the middle of the transform: images that differed from run to run once
            // a second stream loaded the memory system, and wrong images always when requests sat in the M phase's last slots
            // (profiles/r4_wino4_inflight_loads.txt).  Nothing of the epilogue can be scheduled into the loop body.
            if (n + 1 == N) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    W4PROBE(2);
#ifdef CIAOSR_PROBE
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 1024) { g_w4probe[blockIdx.x * 8 + 5] = w4acc5; g_w4probe[blockIdx.x * 8 + 6] = w4acc6; }
#endif
    __syncthreads();                             // every wave is done with V and the patches

    // ---- output transform.  A^T along j in registers: t[r][x]; row half: part[y][x] = sum over this wave's rows A^T[y][3h + r] t[r][x];
    // the wave keeps y = 2h, 2h + 1 and hands y = 2 (1 - h), 2 (1 - h) + 1 to its partner (same nt) through LDS.
    //   xch[w][k = 2 (y & 1) + ... ][q][lane]: k = 4 (y & 1) + x
    float4* xch = reinterpret_cast<float4*>(lds4);
    // Done in four passes over the accumulators' 4-element groups q (the float4 granularity of the exchange and of the stores): a pass
    // holds 18 x 4 accumulator values, 12 t values and its 8 kept float4 -- the whole-tile form kept 12 f32x16 t tiles beside the 18
    // accumulator tiles and spilled 32 registers.
    f32x4 own[4][2][4];                          // [q][yy][x]: part[y = 2h + yy][x], elements 4q .. 4q+3
    auto out_pass = [&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        auto grp = [&](const f32x16& v) __attribute__((always_inline)) -> f32x4 { return __builtin_shufflevector(v, v, 4 * q, 4 * q + 1, 4 * q + 2, 4 * q + 3); };
        f32x4 tt[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const f32x4 m0 = grp(acc[6 * r]), m1 = grp(acc[6 * r + 1]), m2 = grp(acc[6 * r + 2]), m3 = grp(acc[6 * r + 3]), m4 = grp(acc[6 * r + 4]), m5 = grp(acc[6 * r + 5]);
            const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
            tt[r][0] = m0 + s1 + s2;
            tt[r][1] = d1 + 2.f * d2;
            tt[r][2] = s1 + 4.f * s2;
            tt[r][3] = d1 + 8.f * d2 + m5;
        }
        // A^T columns of this wave's rows: h = 0: i = 0, 1, 2 -> y0: 1 1 1, y1: 0 1 -1, y2: 0 1 1, y3: 0 1 -1
        //                                  h = 1: i = 3, 4, 5 -> y0: 1 1 0, y1: 2 -2 0, y2: 4 4 0, y3: 8 -8 1
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            f32x4 keep0, keep1, send0, send1;    // keep y = 2h, 2h + 1; send the other two
            if (h == 0) {
                const f32x4 d = tt[1][x] - tt[2][x], sm = tt[1][x] + tt[2][x];
                keep0 = tt[0][x] + sm; keep1 = d; send0 = sm; send1 = d;
            } else {
                const f32x4 sm = tt[0][x] + tt[1][x], d = tt[0][x] - tt[1][x];
                send0 = sm; send1 = 2.f * d; keep0 = 4.f * sm; keep1 = 8.f * d + tt[2][x];
            }
            own[q][0][x] = keep0;
            own[q][1][x] = keep1;
            xch[((w * 8 + x) * 4 + q) * 64 + lane] = make_float4(send0[0], send0[1], send0[2], send0[3]);
            xch[((w * 8 + 4 + x) * 4 + q) * 64 + lane] = make_float4(send1[0], send1[1], send1[2], send1[3]);
        }
    };
    out_pass(std::integral_constant<int, 0>{});
    out_pass(std::integral_constant<int, 1>{});
    out_pass(std::integral_constant<int, 2>{});
    out_pass(std::integral_constant<int, 3>{});
    __syncthreads();
    W4PROBE(3);
    {
        const int pw = w ^ 2;                    // partner: same nt, other row half
        const __amdgpu_buffer_rsrc_t ors = TABLE ? __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000)
                                                 : __builtin_amdgcn_make_buffer_rsrc(p.x, 0, p.x_bytes, 0x00020000);
        const unsigned o_img = TABLE ? (unsigned)img * p.out_img_bytes : img_off;
        const unsigned o_ld = TABLE ? (unsigned)p.ld_out : (unsigned)p.ldx;
        const int o_col = TABLE ? 64 * (int)blockIdx.z : p.col_out;
        float4 bias4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bias4[q] = TABLE ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(p.bias + 32 * nt + 8 * q + 4 * lh);
        // the partner's halves come back through LDS one (yy, x) ahead of the stores that use them
        float4 o[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[0][q] = xch[((pw * 8 + 0) * 4 + q) * 64 + lane];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int yy = it >> 2, x = it & 3;
            if (it + 1 < 8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[(it + 1) & 1][q] = xch[((pw * 8 + it + 1) * 4 + q) * 64 + lane];
            }
            // The values of this (yy, x) -- one pixel of each of the wave's 32 tiles, 32 channels = 128 B a pixel -- leave in WHOLE lines:
            // written in the accumulator layout (lane = tile) a store instruction covered 32 pixels x 32 B, 32 quarter-used lines (the
            // vector L1's tag rate: the store loop was ~19 % of the kernel over a block's eight layers).  They are turned around in the
            // 4 KB of the partner's exchange slot `it`, whose values this wave took into registers an iteration ago: ds_write_b128 at
            // (tile li, chunk 2 q + lh), chunk-swizzled, ds_read_b128 as (tile 8 j + lane / 8, chunk lane % 8) -> 8 pixels x 128 B a store.
            unsigned char* tb = reinterpret_cast<unsigned char*>(xch + ((pw * 8 + it) * 4) * 64);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 ov = o[it & 1][q];
                float4 v;
                if (TABLE) {                       // plain sums: the table's bias and the softmax come later (head.hip)
                    v.x = own[q][yy][x][0] + ov.x; v.y = own[q][yy][x][1] + ov.y; v.z = own[q][yy][x][2] + ov.z; v.w = own[q][yy][x][3] + ov.w;
                } else {
                    v.x = fmaxf(own[q][yy][x][0] + ov.x + bias4[q].x, 0.f);
                    v.y = fmaxf(own[q][yy][x][1] + ov.y + bias4[q].y, 0.f);
                    v.z = fmaxf(own[q][yy][x][2] + ov.z + bias4[q].z, 0.f);
                    v.w = fmaxf(own[q][yy][x][3] + ov.w + bias4[q].w, 0.f);
                }
                *reinterpret_cast<float4*>(tb + li * 128 + (((2 * q + lh) ^ (li & 7)) << 4)) = v;
            }
            wave_lds_sync();                               // lanes read what OTHER lanes of this wave wrote
            const int sr = lane >> 3, sc = lane & 7;
            const int xx = tx0 + 4 * sr + x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tl = 8 * j + sr;                                    // tile (tyw = j, txw = sr)
                const float4 v = *reinterpret_cast<const float4*>(tb + tl * 128 + ((sc ^ (tl & 7)) << 4));
                const int y = ty0 + 4 * j + 2 * h + yy;
                const bool ok = y < p.H && xx < p.W;
                const unsigned pix = o_img + (unsigned)(y * p.W + xx) * o_ld * 4u + (unsigned)(o_col + 32 * nt + 4 * sc) * 4u;
                i32x4 iv;
                iv.x = __float_as_int(v.x); iv.y = __float_as_int(v.y); iv.z = __float_as_int(v.z); iv.w = __float_as_int(v.w);
                __builtin_amdgcn_raw_buffer_store_b128(iv, ors, (int)(ok ? pix : kOobW4), 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    W4PROBE(4);
#ifdef CIAOSR_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W4PROBE(7);                                  // stores acknowledged
#endif
}

int dense_wino4_tiles(int H, int W) { return ceil_div(H, W4TH) * ceil_div(W, W4TW); }

// dense layer l of a block in Winograd F(4x4, 3x3) form; frag_wino4 = 36 fragment arrays of the transformed weights (encoder_hip.py)
int dense_layer_wino4_f32(float* X, int ldx, int H, int W, int l, const float* frag_wino4, const float* bias, int n_img, hipStream_t s) {
    CIAOSR_CHECK_ARG(X && frag_wino4 && bias && (ldx & 3) == 0 && aligned16(X) && aligned16(frag_wino4) && aligned16(bias));
    const size_t x_bytes = (size_t)n_img * H * W * ldx * 4;
    CIAOSR_CHECK_ARG(n_img >= 1 && n_img <= 65535 && x_bytes < 0xFFFFFF00ull);
    DenseWino4P p;
    p.x = X; p.ldx = ldx; p.x_bytes = (unsigned)x_bytes;
    p.H = H; p.W = W; p.tiles_x = ceil_div(W, W4TW);
    p.groups = l + 1;
    p.wf = reinterpret_cast<const float4*>(frag_wino4);
    p.nj = 64 * (l + 1) / 8;
    p.pos_stride = (long)2 * p.nj * 64;
    p.bias = bias;
    p.col_out = 64 * (l + 1);
    CIAOSR_CHECK_ARG((size_t)36 * p.pos_stride * 16 < 0xFFFFFF00ull);
    p.out = nullptr; p.out_bytes = 0; p.ld_out = 0; p.out_img_bytes = 0; p.wf_block_bytes = 0;
    CIAOSR_BIG_LDS(dense_wino4_f32_kernel<false>, kWino4Lds);
    ProfScope prof("enc_dense_wino4", s);
    hipLaunchKernelGGL(dense_wino4_f32_kernel<false>, dim3(dense_wino4_tiles(H, W), n_img), dim3(256), kWino4Lds, s, p);
    return launch_status("dense_wino4_f32");
}

// Nine 3x3 convolutions 64 -> 64 n_blk channels without bias in F(4x4, 3x3) form: out[(pix * 9 + o) * ldg + n] = sum_{k, c} Pi[o][pix + k][c] w[n][c][k]
// (the logit table of the fused head, head.hip; wino_table_f32 of dense_wino_f32.hip is the F(2x2) form).  Pi: [9][H*W][64]; frag_wino4: 36
// arrays of the transformed [64 n_blk][64] weights in fragment order.
int wino4_table_f32(const float* Pi, int H, int W, const float* frag_wino4, int n_blk, float* out, int ldg, hipStream_t s) {
    CIAOSR_CHECK_ARG(Pi && frag_wino4 && out && n_blk >= 1 && (ldg & 3) == 0 && aligned16(Pi) && aligned16(frag_wino4) && aligned16(out));
    const size_t in_bytes = (size_t)9 * H * W * 64 * 4, out_bytes = (size_t)9 * H * W * ldg * 4;
    CIAOSR_CHECK_ARG(in_bytes < 0xFFFFFF00ull && out_bytes < 0xFFFFFF00ull);
    DenseWino4P p;
    p.x = const_cast<float*>(Pi); p.ldx = 64; p.x_bytes = (unsigned)in_bytes;       // the nine maps = nine "images" of one 64-channel group
    p.H = H; p.W = W; p.tiles_x = ceil_div(W, W4TW);
    p.groups = 1;
    p.wf = reinterpret_cast<const float4*>(frag_wino4);
    p.nj = 8;
    p.pos_stride = (long)2 * n_blk * p.nj * 64;                                      // float4 per position: 2 n_blk channel halves x 8 k-chunks x 64 lanes
    p.wf_block_bytes = 2u * 8u * 1024u;                                              // one 64-channel block = two halves
    p.bias = nullptr; p.col_out = 0;
    p.out = out; p.out_bytes = (unsigned)out_bytes; p.ld_out = 9 * ldg; p.out_img_bytes = (unsigned)ldg * 4u;
    CIAOSR_CHECK_ARG((size_t)36 * p.pos_stride * 16 < 0xFFFFFF00ull);
    CIAOSR_BIG_LDS(dense_wino4_f32_kernel<true>, kWino4Lds);
    ProfScope prof("head_logit_table_w4", s);   // F(4x4) form: 36 / 144 of the direct form's MFMAs (bench.py executed_ratio)
    hipLaunchKernelGGL(dense_wino4_f32_kernel<true>, dim3(dense_wino4_tiles(H, W), 9, n_blk), dim3(256), kWino4Lds, s, p);
    return launch_status("wino4_table_f32");
}

}  // namespace ciaosr

#ifdef CIAOSR_PROBE
extern "C" int ciaosr_debug_probe_w4_read(unsigned long long* host, int n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ciaosr::g_w4probe), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
#endif
