// Implicit-GEMM convolution (1x1 / 3x3, stride 1, zero 'same' padding) on channels-last maps,
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   dst[p][co] = act((sum_{tap,ci} src[p + tap][ci] * w[co][tap*Cin + ci] + bias[co]) * alpha) + res[p][co]
//
// The A operand is never materialised: a K-stage of 64 input channels is gathered from the (shifted)
// channels-last map, 256 contiguous bytes per pixel.  Dense-block concatenation (RDN) is free: layers write
// their output channels into the next columns of one [HW][C_total] buffer.
//
// Workgroup = 8 waves on a TM x TN output tile (TM = TN in {32, 64}); the NT = TM*TN/1024 MFMA tiles are
// each computed by 8/NT waves that split every 64-deep K-stage between them (all four SIMDs work on the
// tile, two waves per SIMD), and the K-slices are summed through LDS at the end.  A 48x48 image has only
// 2304 x 64 outputs per layer: small tiles (32x32 -> 1152 workgroups for a 512-wide dense-block step) spread
// the layer over every CU and shorten each workgroup's dependent chain of stages; big images use 64x64.
// When even that leaves CUs idle the K loop is split over blockIdx.y and a second kernel reduces the partial
// slabs in a fixed order (deterministic) and applies the epilogue.
//
// Memory accesses that may fall outside the image / matrix go through buffer descriptors: an out-of-range
// offset makes a load return 0 and a store vanish, with no branch.  (A "cond ? load : 0" form makes hipcc
// branch around each load and wait vmcnt(0) per element: the 4 loads of a stage, or the 16 read-modify-writes
// of an epilogue, become dependent memory round trips -- measured 2.6 us per stage / 10 us per epilogue.)
//
// "Scatter form" of a residual dense block: input group s (64 channels) is convolved once with the stacked
// weight slices of ALL later dense layers (N = 64*(L-s), K = 576) and accumulated into running sums, instead
// of L convolutions with growing K = 576*(l+1) and N = 64 -- same MACs, L launches with wide N instead of
// 2L launches (conv + split-K reduce) with a 36-tile grid.
//
// Replaces the encoder trunk convolutions the reference runs through torch conv2d:
// RDN / EDSR `gen_feature` (ciaosr_net.py:321-342, :393-408).

#include "ops.h"

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int CBK = 64, CLD = CBK + 4;
constexpr unsigned kOob = 0xFFFFFFF0u;

struct ConvP {
    const float* src; int ld_src; int H, W, Cin;
    const float* wgt; int ldw;
    unsigned src_bytes, wgt_bytes;   // extents for the buffer descriptors (< 4 GiB)
    const float* bias;
    int Cout, taps;
    float* dst; int ld_dst;
    float* dst2; int ld_dst2;
    const float* res; int ld_res;
    int act; float alpha;
    int M, K;
    int tiles_n, splitk, kt_per_split;
    float* partial;                  // [splitk][M][Cout] when splitk > 1
    // dense-block scatter mode (dense_step >= 0): 64-column block nt of the output belongs to layer dense_step + nt
    int dense_step;
    float* acc_buf; int ld_acc;      // [M][64 * num_layers] running pre-activation sums
    const float* dense_bias;         // [num_layers][64]
};

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

// scalar epilogue (split-K reduce kernel: one element per thread)
__device__ __forceinline__ void conv_epilogue(const ConvP& p, int row, int col, float v) {
    if (p.dense_step >= 0) {
        const int nt = col >> 6, c = col & 63, l = p.dense_step + nt;
        float* a = p.acc_buf + (size_t)row * p.ld_acc + 64 * l + c;
        if (p.dense_step > 0) v += *a;
        if (nt == 0)
            p.dst[(size_t)row * p.ld_dst + 64 * (p.dense_step + 1) + c] = fmaxf(v + p.dense_bias[64 * l + c], 0.f);
        else
            *a = v;
        return;
    }
    v = (v + (p.bias ? p.bias[col] : 0.f)) * p.alpha;
    if (p.act == CIAOSR_ACT_RELU) v = fmaxf(v, 0.f);
    else if (p.act == CIAOSR_ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    if (p.res) v += p.res[(size_t)row * p.ld_res + col];
    p.dst[(size_t)row * p.ld_dst + col] = v;
    if (p.dst2) p.dst2[(size_t)row * p.ld_dst2 + col] = v;
}

// ---- tile epilogue: 16 independent buffer accesses per lane (see the header comment).  rrow[r] = global output
// row (pixel) of accumulator register r or 0xFFFFFFFF; col = this lane's output channel, col0 = the tile's first.
__device__ __forceinline__ void conv_tile_epilogue(const ConvP& p, f32x16& acc, const unsigned (&rrow)[16], int col, int col0,
                                                   bool cok, int split) {
    auto off = [&](int r, int ld, int c) -> int {
        return rrow[r] == 0xFFFFFFFFu ? (int)kOob : (int)((rrow[r] * (unsigned)ld + (unsigned)c) * 4u);
    };
    if (p.splitk > 1) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            p.partial + (size_t)split * p.M * p.Cout, 0, (unsigned)((size_t)p.M * p.Cout * 4), 0x00020000);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(acc[r]), rs, off(r, p.Cout, col), 0, 0);
        return;
    }
    if (p.dense_step >= 0) {
        const int nt = col0 >> 6;                              // wave-uniform: a 32-wide tile lies in one layer
        const int c = col & 63, l = p.dense_step + nt;
        const __amdgpu_buffer_rsrc_t rs_acc =
            __builtin_amdgcn_make_buffer_rsrc(p.acc_buf, 0, (unsigned)((size_t)p.M * p.ld_acc * 4), 0x00020000);
        if (p.dense_step > 0) {
            float prev[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                prev[r] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_acc, off(r, p.ld_acc, 64 * l + c), 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += prev[r];
        }
        if (nt == 0) {
            const __amdgpu_buffer_rsrc_t rs_x =
                __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, (unsigned)((size_t)p.M * p.ld_dst * 4), 0x00020000);
            const float b = cok ? p.dense_bias[64 * l + c] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(fmaxf(acc[r] + b, 0.f)), rs_x,
                                                      off(r, p.ld_dst, 64 * (p.dense_step + 1) + c), 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(acc[r]), rs_acc, off(r, p.ld_acc, 64 * l + c), 0, 0);
        }
        return;
    }
    {
        const float b = (p.bias && cok) ? p.bias[col] : 0.f;
        float resv[16];
        if (p.res) {
            const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.res), 0, (unsigned)(((size_t)(p.M - 1) * p.ld_res + p.Cout) * 4), 0x00020000);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                resv[r] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_r, off(r, p.ld_res, col), 0, 0));
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) resv[r] = 0.f;
        }
        const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(
            p.dst, 0, (unsigned)(((size_t)(p.M - 1) * p.ld_dst + p.Cout) * 4), 0x00020000);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = (acc[r] + b) * p.alpha;
            if (p.act == CIAOSR_ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == CIAOSR_ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
            v += resv[r];
            acc[r] = v;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rs_d, off(r, p.ld_dst, col), 0, 0);
        }
        if (p.dst2) {
            const __amdgpu_buffer_rsrc_t rs_2 = __builtin_amdgcn_make_buffer_rsrc(
                p.dst2, 0, (unsigned)(((size_t)(p.M - 1) * p.ld_dst2 + p.Cout) * 4), 0x00020000);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(acc[r]), rs_2, off(r, p.ld_dst2, col), 0, 0);
        }
    }
}

template <int TM, int TN>
struct ConvCfg {
    static constexpr int NT = (TM / 32) * (TN / 32);   // MFMA tiles per workgroup: 1, 2 or 4
    static constexpr int KS = 8 / NT;                  // K-slices per stage (waves per tile): 8, 4 or 2
    static constexpr int JW = 8 / KS;                  // 8-deep k-chunks per wave per stage: 1, 2 or 4
    static constexpr int RA = TM / 32, RB = TN / 32;   // float4 staging loads per thread (A rows, B rows)
    static constexpr size_t lds = (size_t)2 * (TM + TN) * CLD * sizeof(float);
};

template <int TM, int TN>
__global__ __launch_bounds__(512) void conv_gemm_kernel(ConvP p) {
    using Cfg = ConvCfg<TM, TN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][TM * CLD]
    float* Bs = smem + 2 * TM * CLD;           // [2][TN * CLD]
    const int tile = blockIdx.x;
    const int m0 = (tile / p.tiles_n) * TM, n0 = (tile % p.tiles_n) * TN;
    const int split = blockIdx.y;
    const int nst_total = (p.K + CBK - 1) / CBK;
    const int st0 = split * p.kt_per_split;
    const int nst = min(nst_total, st0 + p.kt_per_split) - st0;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int tl = w % Cfg::NT, ks = w / Cfg::NT;            // MFMA tile and K-slice of this wave
    const int wm = tl / (TN / 32), wn = tl % (TN / 32);

    // staging: thread -> float4 column c4 of rows r0 + 32*s
    const int r0 = t >> 4, c4 = (t & 15) * 4;
    int py[Cfg::RA], px[Cfg::RA];
    bool pv[Cfg::RA];
#pragma unroll
    for (int s = 0; s < Cfg::RA; ++s) {
        const int m = m0 + r0 + 32 * s;
        pv[s] = m < p.M;
        py[s] = pv[s] ? m / p.W : 0;
        px[s] = pv[s] ? m - py[s] * p.W : 0;
    }
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, p.wgt_bytes, 0x00020000);

    struct Regs {
        float4 a[Cfg::RA], b[Cfg::RB];
    };
    auto load_stage = [&](int st, Regs& R) {
        const int k = st * CBK + c4;
        const bool kok = k < p.K;
        const int tap = kok ? k / p.Cin : 0, cc = k - tap * p.Cin;
        const int dy = p.taps == 9 ? tap / 3 - 1 : 0, dx = p.taps == 9 ? tap % 3 - 1 : 0;
#pragma unroll
        for (int s = 0; s < Cfg::RA; ++s) {
            const int y = py[s] + dy, x = px[s] + dx;
            const bool ok = kok && pv[s] && y >= 0 && y < p.H && x >= 0 && x < p.W;
            R.a[s] = buf_load4(rs_a, ok ? ((unsigned)(y * p.W + x) * (unsigned)p.ld_src + (unsigned)cc) * 4u : kOob);
        }
#pragma unroll
        for (int s = 0; s < Cfg::RB; ++s) {
            const int n = n0 + r0 + 32 * s;
            R.b[s] = buf_load4(rs_b, (kok && n < p.Cout) ? ((unsigned)n * (unsigned)p.ldw + (unsigned)k) * 4u : kOob);
        }
    };
    auto store_stage = [&](int buf, const Regs& R) {
#pragma unroll
        for (int s = 0; s < Cfg::RA; ++s)
            *reinterpret_cast<float4*>(As + buf * TM * CLD + (r0 + 32 * s) * CLD + c4) = R.a[s];
#pragma unroll
        for (int s = 0; s < Cfg::RB; ++s)
            *reinterpret_cast<float4*>(Bs + buf * TN * CLD + (r0 + 32 * s) * CLD + c4) = R.b[s];
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto compute = [&](int buf) {
        const float* a = As + buf * TM * CLD + (wm * 32 + li) * CLD + 8 * Cfg::JW * ks + 4 * lh;
        const float* b = Bs + buf * TN * CLD + (wn * 32 + li) * CLD + 8 * Cfg::JW * ks + 4 * lh;
#pragma unroll
        for (int j = 0; j < Cfg::JW; ++j) {
            const float4 fa = *reinterpret_cast<const float4*>(a + 8 * j);
            const float4 fb = *reinterpret_cast<const float4*>(b + 8 * j);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
        }
    };

    // two register sets / two LDS buffers: iteration i computes stage i from LDS, stores stage i+1 (loaded one
    // iteration ago) into the other buffer and issues the loads of stage i+2
    Regs R0, R1;
    if (nst > 0) load_stage(st0, R0);
    if (nst > 1) load_stage(st0 + 1, R1);
    if (nst > 0) store_stage(0, R0);
    __syncthreads();
    for (int i = 0; i < nst; i += 2) {
        if (i + 2 < nst) load_stage(st0 + i + 2, R0);
        compute(0);
        if (i + 1 < nst) store_stage(1, R1);
        __syncthreads();
        if (i + 1 >= nst) break;
        if (i + 3 < nst) load_stage(st0 + i + 3, R1);
        compute(1);
        if (i + 2 < nst) store_stage(0, R0);
        __syncthreads();
    }

    // sum the K-slices through LDS (the stage buffers are free now): slice 0 of every tile collects
    if constexpr (Cfg::KS > 1) {
        float* red = smem;   // [(KS-1)*NT][16][64]
        if (ks > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((ks - 1) * Cfg::NT + tl) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int s = 1; s < Cfg::KS; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[(((s - 1) * Cfg::NT + tl) * 16 + r) * 64 + lane];
    }

    const int col = n0 + wn * 32 + li;
    const bool cok = col < p.Cout;
    unsigned rrow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        rrow[r] = (cok && row < p.M) ? (unsigned)row : 0xFFFFFFFFu;
    }
    conv_tile_epilogue(p, acc, rrow, col, n0 + wn * 32, cok, split);
}

// ---- 3x3 convolution with a halo-resident A tile ------------------------------------------------------
// The per-CU global-load path sustains only ~10-15 B/clk on this chip (PMC + timing of the tap-by-tap kernel
// above: a 48x48 layer takes 12-26 us against an 8.6 us MFMA bound while L2 hits 92 %), and re-reading the
// pixels once per tap is 9x redundant.  Here a workgroup owns an 8x8 pixel block x 32 output channels: the
// 10x10-pixel halo of 64 input channels (27 KB) is loaded into LDS ONCE and all nine taps read shifted rows of it;
// only the weights (8 KB per tap) are streamed.  Bytes per output drop from 141 (32x32 tile) to 48 and the
// layer becomes MFMA-bound even at 48x48.  8 waves = 2 MFMA tiles (pixels 0-31 / 32-63 of the block) x 4 slices of
// the 64-deep tap; K-slices are summed through LDS.  Channel groups of 64 (Cin = 64 g) reload the halo per group.
#ifdef CIAOSR_PROBE
__device__ unsigned long long g_probe[4096 * 8];
#define PROBE(slot) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) g_probe[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define PROBE(slot) do { } while (0)
#endif

constexpr int HT = 8, HHALO = HT + 2;                 // 8x8 pixels, 10x10 halo
constexpr int HA_FLOATS = HHALO * HHALO * CLD;         // 6800
constexpr int HBK = 3 * 64, HBLD = HBK + 4;            // a stage = one row of the 3x3 kernel (3 taps x 64 channels)
constexpr int HB_FLOATS = 32 * HBLD;                   // 6272 per stage
constexpr size_t kHaloLds = (size_t)(HA_FLOATS + 2 * HB_FLOATS) * sizeof(float);   // 77 376 B: two workgroups per CU

__global__ __launch_bounds__(512) void conv3x3_halo_kernel(ConvP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ah = smem;                       // [100][CLD]
    float* Bs = smem + HA_FLOATS;           // [2][32][HBLD]
    const int tiles_x = (p.W + HT - 1) / HT;
    const int tile = blockIdx.x;
    const int mt = tile / p.tiles_n, n0 = (tile % p.tiles_n) * 32;
    const int ty0 = (mt / tiles_x) * HT, tx0 = (mt % tiles_x) * HT;
    const int split = blockIdx.y;
    const int nst_total = 3 * (p.Cin / 64);  // stages: (channel group, kernel row)
    const int st0 = split * p.kt_per_split;
    const int nst = min(nst_total, st0 + p.kt_per_split) - st0;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int tl = w & 1, ks = w >> 1;                         // MFMA tile (pixel half) and K-slice (16 channels)
    const int m = 32 * tl + li;                                // pixel of this lane inside the 8x8 block
    const int hbase = ((m >> 3) + 1) * HHALO + (m & 7) + 1;    // its halo index for tap (0,0)

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, p.wgt_bytes, 0x00020000);

    // weight stage: 32 output channels x 3 taps x 64 channels = 1536 float4, three per thread.
    // k index of (stage st = 3*cg + krow, tap-in-row j, channel c): (3*krow + j)*Cin + 64*cg + c
    auto load_b = [&](int st, float4 (&rb)[3]) {
        const int cg = st / 3, krow = st - 3 * cg;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int idx = t + 512 * e;                       // 0..1535
            const int r = idx / 48, q = idx - 48 * r;          // row (output channel), float4 column 0..47
            const int j = q >> 4, c4 = (q & 15) * 4;
            const int n = n0 + r;
            rb[e] = buf_load4(rs_b, n < p.Cout ? ((unsigned)n * (unsigned)p.ldw + (unsigned)((3 * krow + j) * p.Cin + 64 * cg + c4)) * 4u : kOob);
        }
    };
    auto store_b = [&](int buf, const float4 (&rb)[3]) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int idx = t + 512 * e;
            const int r = idx / 48, q = idx - 48 * r;
            *reinterpret_cast<float4*>(Bs + buf * HB_FLOATS + r * HBLD + 4 * q) = rb[e];
        }
    };
    auto load_halo = [&](int cg) {
        for (int idx = t; idx < HHALO * HHALO * 16; idx += 512) {
            const int hp = idx >> 4, c4 = (idx & 15) * 4;
            const int y = ty0 - 1 + hp / HHALO, x = tx0 - 1 + hp % HHALO;
            const bool ok = y >= 0 && y < p.H && x >= 0 && x < p.W;
            const float4 v = buf_load4(rs_a, ok ? ((unsigned)(y * p.W + x) * (unsigned)p.ld_src + (unsigned)(64 * cg + c4)) * 4u : kOob);
            *reinterpret_cast<float4*>(Ah + hp * CLD + c4) = v;
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    int cur_cg = -1;
    float4 rb[3];
    PROBE(0);
    if (nst > 0) load_b(st0, rb);
    for (int i = 0; i < nst; ++i) {
        const int st = st0 + i;
        const int cg = st / 3, krow = st - 3 * cg;
        if (i == 1) PROBE(3);
        if (cg != cur_cg) {                  // (block-uniform) new channel group: refill the halo tile
            __syncthreads();                 // every wave is done with the previous group's halo
            load_halo(cg);
            cur_cg = cg;
        }
        if (i == 0) PROBE(1);
        store_b(i & 1, rb);
        if (i + 1 < nst) load_b(st + 1, rb);
        __syncthreads();
        if (i == 0) PROBE(2);
        const float* a = Ah + (hbase + (krow - 1) * HHALO - 1) * CLD + 16 * ks + 4 * lh;   // tap (krow-1, -1)
        const float* b = Bs + (i & 1) * HB_FLOATS + li * HBLD + 16 * ks + 4 * lh;
#pragma unroll
        for (int j = 0; j < 3; ++j) {        // the three taps of this kernel row: dx = -1, 0, +1
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 fa = *reinterpret_cast<const float4*>(a + j * CLD + 8 * h);
                const float4 fb = *reinterpret_cast<const float4*>(b + 64 * j + 8 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
            }
        }
    }
    PROBE(4);
    __syncthreads();

    // sum the 4 K-slices of each MFMA tile through LDS (halo / weight stages are free now)
    float* red = smem;   // [3*2][16][64] = 6144 floats <= HA_FLOATS
    if (ks > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((ks - 1) * 2 + tl) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll
    for (int s = 1; s < 4; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(((s - 1) * 2 + tl) * 16 + r) * 64 + lane];

    const int col = n0 + li;
    const bool cok = col < p.Cout;
    unsigned rrow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mm = 32 * tl + (r & 3) + 8 * (r >> 2) + 4 * lh;       // pixel of this accumulator register
        const int y = ty0 + (mm >> 3), x = tx0 + (mm & 7);
        rrow[r] = (cok && y < p.H && x < p.W) ? (unsigned)(y * p.W + x) : 0xFFFFFFFFu;
    }
    PROBE(5);
    conv_tile_epilogue(p, acc, rrow, col, n0, cok, split);
    PROBE(6);
}

__global__ void conv_reduce_kernel(ConvP p) {
    const long n = (long)p.M * p.Cout;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += p.partial[(size_t)s * n + i];   // fixed order: deterministic
        const int row = (int)(i / p.Cout), col = (int)(i - (long)row * p.Cout);
        conv_epilogue(p, row, col, v);
    }
}

// ---- launch policy ------------------------------------------------------------------------------
template <int TM, int TN>
static int launch_tile(const ConvP& p, int tiles, hipStream_t s) {
    constexpr size_t lds = ConvCfg<TM, TN>::lds;
    auto kernel = conv_gemm_kernel<TM, TN>;
    CIAOSR_BIG_LDS(kernel, lds);
    hipLaunchKernelGGL(kernel, dim3(tiles, p.splitk), dim3(512), lds, s, p);
    return CIAOSR_OK;
}

// picks the tile shape and the K split, launches the GEMM (+ the slab reduce); p.* geometry already filled
static int launch_conv(ConvP& p, float* partial, size_t partial_floats, hipStream_t s, const char* tag) {
    constexpr int target_wg = 512;
    const long halo_tiles = (long)ceil_div(p.H, HT) * ceil_div(p.W, HT) * ceil_div(p.Cout, 32);
    if (p.taps == 9 && (p.Cin & 63) == 0 && halo_tiles <= 4096) {
        // small 3x3 layers: halo-resident A tile, 8x8 pixels x 32 output channels per workgroup (the per-workgroup
        // fixed costs -- halo fill, K-slice reduction, read-modify-write epilogue: ~55 % of its cycles by the
        // in-kernel probe -- make the plain 64x64 tap kernel the faster one on big images, where tiles abound)
        const int nst9 = 3 * (p.Cin / 64);      // stages of one kernel row (3 taps x 64 channels)
        p.tiles_n = ceil_div(p.Cout, 32);
        const int tiles = (int)halo_tiles;
        int splitk = 1;
        if (partial && tiles < 128) {
            splitk = ceil_div(256, tiles);
            if (splitk > nst9) splitk = nst9;
            if (splitk > 16) splitk = 16;
            const size_t per = (size_t)p.M * p.Cout;
            if ((size_t)splitk * per > partial_floats) splitk = (int)(partial_floats / per);
            if (splitk < 1) splitk = 1;
        }
        p.kt_per_split = ceil_div(nst9, splitk);
        p.splitk = ceil_div(nst9, p.kt_per_split);
        p.partial = partial;
        {
            ProfScope prof(tag, s);
            CIAOSR_BIG_LDS(conv3x3_halo_kernel, kHaloLds);
            hipLaunchKernelGGL(conv3x3_halo_kernel, dim3(tiles, p.splitk), dim3(512), kHaloLds, s, p);
        }
        int rc = launch_status("conv3x3_halo");
        if (rc != CIAOSR_OK) return rc;
        if (p.splitk > 1) {
            ProfScope prof("conv_splitk_reduce", s);
            const long n = (long)p.M * p.Cout;
            int grid = (int)((n + 255) / 256);
            hipLaunchKernelGGL(conv_reduce_kernel, dim3(grid > 2048 ? 2048 : grid), dim3(256), 0, s, p);
            rc = launch_status("conv_reduce");
        }
        return rc;
    }
    const int nst = (p.K + CBK - 1) / CBK;
    const long out32 = (long)ceil_div(p.M, 32) * ceil_div(p.Cout, 32);
    // small output (a 48x48 layer): 32x32 tiles spread it over every CU; large output: 64x64 tiles halve the
    // operand traffic per MAC
    // (a tall narrow output -- the 1x1 LFF of a 192x192 tile, M = 36 864, Cout = 64 -- has few 32x32 tiles per column but plenty
    // of rows: 64x64 tiles there too)
    const int tm = (out32 <= 4096 && p.M < 16384) ? 32 : 64;        // (128x64, one workgroup per CU, measured 7 % slower than 64x64 at the 192 tile)
    if (p.dense_step >= 0 && (p.Cout & 63)) return CIAOSR_ERR_BAD_ARG;
    const int tn = tm;
    p.tiles_n = ceil_div(p.Cout, tn);
    const int tiles = ceil_div(p.M, tm) * p.tiles_n;
    int splitk = 1;
    if (partial && tiles < target_wg / 2) {
        splitk = ceil_div(target_wg, tiles);
        if (splitk > nst) splitk = nst;
        if (splitk > 16) splitk = 16;
        const size_t per = (size_t)p.M * p.Cout;
        if ((size_t)splitk * per > partial_floats) splitk = (int)(partial_floats / per);
        if (splitk < 1) splitk = 1;
    }
    p.kt_per_split = ceil_div(nst, splitk);
    p.splitk = ceil_div(nst, p.kt_per_split);
    p.partial = partial;
    {
        ProfScope prof(tag, s);
        const int rc_attr = tm == 32 ? launch_tile<32, 32>(p, tiles, s) : launch_tile<64, 64>(p, tiles, s);
        if (rc_attr != CIAOSR_OK) return rc_attr;
    }
    int rc = launch_status("conv_gemm");
    if (rc != CIAOSR_OK) return rc;
    if (p.splitk > 1) {
        ProfScope prof("conv_splitk_reduce", s);
        const long n = (long)p.M * p.Cout;
        int grid = (int)((n + 255) / 256);
        hipLaunchKernelGGL(conv_reduce_kernel, dim3(grid > 2048 ? 2048 : grid), dim3(256), 0, s, p);
        rc = launch_status("conv_reduce");
    }
    return rc;
}

// src [H*W][ld_src] (first Cin columns) -> dst [H*W][ld_dst] (first Cout columns)
int conv2d_hwc(const float* src, int ld_src, int H, int W, int Cin, const float* wgt, int ldw, const float* bias,
               int Cout, int ksize, float* dst, int ld_dst, float* dst2, int ld_dst2, const float* res, int ld_res,
               int act, float alpha, float* partial, size_t partial_floats, hipStream_t s, const char* tag) {
    CIAOSR_CHECK_ARG(src && wgt && dst && (ksize == 1 || ksize == 3));
    CIAOSR_CHECK_ARG(Cin % 32 == 0 && (ld_src & 3) == 0 && (ldw & 3) == 0);
    CIAOSR_CHECK_ARG(aligned16(src) && aligned16(wgt));
    ConvP p;
    p.src = src; p.ld_src = ld_src; p.H = H; p.W = W; p.Cin = Cin;
    p.wgt = wgt; p.ldw = ldw; p.bias = bias; p.Cout = Cout; p.taps = ksize * ksize;
    p.dst = dst; p.ld_dst = ld_dst; p.dst2 = dst2; p.ld_dst2 = ld_dst2; p.res = res; p.ld_res = ld_res;
    p.act = act; p.alpha = alpha;
    p.M = H * W; p.K = p.taps * Cin;
    p.dense_step = -1; p.acc_buf = nullptr; p.ld_acc = 0; p.dense_bias = nullptr;
    const size_t sb = ((size_t)(p.M - 1) * ld_src + Cin) * sizeof(float), wb = ((size_t)(Cout - 1) * ldw + p.K) * sizeof(float);
    CIAOSR_CHECK_ARG(sb < 0xFFFFFF00ull && wb < 0xFFFFFF00ull);   // 32-bit buffer offsets
    p.src_bytes = (unsigned)sb; p.wgt_bytes = (unsigned)wb;
    return launch_conv(p, partial, partial_floats, s, tag ? tag : (ksize == 3 ? "conv3x3" : "conv1x1"));
}

// One scatter step of a residual dense block (all layers 64 wide): X[:, 64*step : 64*step+64] (3x3, zero
// pad) -> contributions to dense layers step .. num_layers-1.  wgt: [64*(num_layers-step)][9*64] stacked
// weight slices; completes layer `step` into X[:, 64*(step+1) ...].
int dense_scatter_step(float* X, int ldx, int H, int W, int step, int num_layers, const float* wgt, const float* bias_all,
                       float* acc_buf, float* partial, size_t partial_floats, hipStream_t s) {
    ConvP p;
    p.src = X + 64 * step; p.ld_src = ldx; p.H = H; p.W = W; p.Cin = 64;
    p.wgt = wgt; p.ldw = 9 * 64; p.bias = nullptr; p.Cout = 64 * (num_layers - step); p.taps = 9;
    p.dst = X; p.ld_dst = ldx; p.dst2 = nullptr; p.ld_dst2 = 0; p.res = nullptr; p.ld_res = 0;
    p.act = CIAOSR_ACT_NONE; p.alpha = 1.f;
    p.M = H * W; p.K = 9 * 64;
    p.dense_step = step; p.acc_buf = acc_buf; p.ld_acc = 64 * num_layers; p.dense_bias = bias_all;
    const size_t sb = ((size_t)(p.M - 1) * ldx + 64) * sizeof(float), wb = (size_t)p.Cout * p.ldw * sizeof(float);
    CIAOSR_CHECK_ARG(sb < 0xFFFFFF00ull && wb < 0xFFFFFF00ull && (size_t)p.M * ldx * 4 < 0xFFFFFF00ull);
    p.src_bytes = (unsigned)sb; p.wgt_bytes = (unsigned)wb;
    return launch_conv(p, partial, partial_floats, s, "enc_dense_scatter");
}

}  // namespace ciaosr
