// fp32 dense-block convolution for big maps in WINOGRAD form, F(2x2, 3x3) (round 3): the same 3x3 dense layer as dense_f32.hip
// (mmedit RDB.layers[l].conv over cat(x, d_0 .. d_{l-1}), called from ciaosr_net.py:330-337) with 16 multiplies per 2x2 output tile and
// (ci, co) pair instead of 36: 2.25x fewer MFMAs, and -- because a workgroup's 128 pixels are exactly one 32-row MFMA tile of
// Winograd tiles -- none of the 160-for-144 row padding of the direct kernel (2.8x fewer MFMA cycles per pixel and input group).
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A        g 3x3 weights, d the 4x4 input tile whose top-left is the output tile's (-1, -1)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]      G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]      A^T = [1 1 1 0; 0 1 -1 -1]
// fp32 throughout on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32); the transforms are additions (B, A) and host-side
// pre-multiplication of the weights (G, in fp64, rounded once).  Not bitwise a direct fmaf chain: the products are of transformed
// operands (error growth of F(2x2, 3x3) ~ 4x a direct convolution's rounding, measured against the oracle in the tests).
//
// Decomposition:
//   * workgroup = 8 x 16 output pixels = 4 x 8 Winograd tiles = the 32 rows of one MFMA tile, x all 64 output channels;
//   * 8 waves = 4 rows i of the 4x4 transformed domain x 2 output-channel halves: wave (i, nt) owns positions (i, 0..3) of its 32
//     channels = 4 accumulator tiles (64 registers; two waves per SIMD); every wave runs the FULL K loop for its positions, so there is no K-slice reduction -- the only
//     cross-wave step is the row half of the output transform (A^T along i), 64 KB through LDS once per layer;
//   * the 10 x 18-pixel halo patch of one 64-channel input group (fp32, 46 KB, unpadded 256-B pixels whose sixteen 16-B chunks are
//     XOR-swizzled with ((px >> 1) & 7) | (((py >> 1) & 1) << 3): the 16 lanes of a ds_read_b128 group -- tiles two pixels apart --
//     hit 16 distinct bank quads for every (a, b) of the 4x4 window) is staged once per group by LDS-DMA into the other of two buffers;
//   * per 8-channel step a wave reads the 2 x 4 window pixels its row needs (8 ds_read_b128), forms its four transformed values
//     with 32 additions, and issues 16 MFMAs against 4 pre-packed 1-KB weight fragments from L2 (the transformed weights
//     U[p] = (G g G^T)[p] as 16 separate [64][cin] matrices in ciaosr_pack_fragments_f32 order).
//
// Round 4, measured and NOT kept: 4 waves per workgroup (one per SIMD), each owning BOTH channel halves of its transformed row (8
// accumulator tiles), so that the input transform is computed once per row instead of by both channel-half waves (1.3 instead of 2.6
// VALU per MFMA).  Correct (same tests), no spills (256 VGPR + 133 AGPR), and SLOWER on the same box: enc_dense_wino 753 -> 783 ms per
// C3 step, logit table 61.5 -> 63.8.  It is what tools/ubench/mfma_valu.hip predicts: with ONE wave per SIMD a VALU instruction beside
// an fp32 MFMA costs ~20 cycles of the pipe (64 -> 84 at one per MFMA), with two waves ~6 (64 -> 70): 1.3 VALU per MFMA alone on a
// SIMD is ~86 cycles per MFMA, 2.6 shared by two waves ~81.  Halving the VALU work only pays with two waves per SIMD, and two waves
// of 8 accumulator tiles + a weight ring do not fit 256 registers each.
#include <cstdlib>

#include "ops.h"

namespace ciaosr {

#ifdef CIAOSR_PROBE      // developer probe build (make probe; tools/wino_probe.py): per-wave stamps of the launches with g_wprobe_groups input groups
__device__ unsigned long long g_wprobe[1024 * 8 * 32];
__device__ int g_wprobe_groups = 2;
#define WPROBE(slot) do { if (!TABLE && lane == 0 && blockIdx.y == gridDim.y / 2 && blockIdx.x < 1024 && p.groups == g_wprobe_groups) \
        g_wprobe[(blockIdx.x * 8 + wv) * 32 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define WPROBE(slot) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int WTH = 8, WTW = 16;                 // output tile (pixels)
constexpr int WPH = WTH + 2, WPW = WTW + 2;      // patch with the 1-pixel halo
constexpr int WPATCH = WPH * WPW * 256;          // 46 080 B per buffer (64 fp32 per pixel, unpadded, chunk-swizzled)
constexpr int WCHUNKS = WPH * WPW * 16;
constexpr size_t kWinoLds = 2 * (size_t)WPATCH;  // 92 160 B >= the 64-KB output-transform scratch
constexpr unsigned kOobW = 0xFFFFFFF0u;

struct DenseWinoP {
    float* x; int ldx;                           // output map [H*W][ldx] (dense layers: also the input)
    const float* in; int ld_in;                  // input maps, back to back: image (dense) / key offset (logit table) blockIdx.y
    unsigned x_bytes;                            // extent of `in`
    int n_blk;                                   // 64-channel output blocks per workgroup (dense layers: 1; logit table: 4)
    int out_y_stride;                            // floats added to the output address per blockIdx.y within a pixel row (logit table: ldg)
    int H, W, tiles_x;
    int groups;
    const float4* wf;                            // 16 fragment arrays [2][nj][64 lanes] float4, one per transformed position, back to back
    int nj; long pos_stride;                     // nj = cin / 8; float4 per position array
    const float* bias;
    int col_out;
};

__device__ __forceinline__ int wino_swz(int py, int px) { return ((px >> 1) & 7) | (((py >> 1) & 1) << 3); }

// TABLE = false: a dense layer (input = output buffer, bias + ReLU, one output block, blockIdx.y = image).
// TABLE = true : the logit table of the fused head as nine 3x3 convolutions 64 -> 256 (head.hip): blockIdx.y = key offset o, input map
//                Pi_o, four output blocks from the ONE resident patch (the output-transform scratch sits behind it), no bias, no ReLU.
template <bool TABLE>
__global__ __launch_bounds__(512, 2) void dense_wino_f32_kernel(DenseWinoP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsw[];
    // 8 waves = 4 rows i of the transformed domain x 2 output-channel halves: two waves per SIMD cover each other's LDS round trips,
    // transform VALU and weight waits (the window reads and the transform are done by both waves of a row: LDS and VALU have the room)
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, lh = lane >> 5;
    const int w = wv & 3, nth = wv >> 2;
    const int ty0 = (blockIdx.x / p.tiles_x) * WTH, tx0 = (blockIdx.x % p.tiles_x) * WTW;
    const int img = blockIdx.y;
    WPROBE(0);
    const int blk0 = TABLE ? (int)blockIdx.z * p.n_blk : 0;     // logit table: first of this workgroup's 64-channel output blocks
    const unsigned img_off = (unsigned)((size_t)img * p.H * p.W * p.ld_in * 4);

    // patch staging by LDS-DMA (`buffer_load_dwordx4 ... lds`, no staging registers): piece i = wv + 8 s (i < 45) fills LDS bytes
    // [1024 i, 1024 i + 1024) of a patch buffer; lane -> LDS chunk c = 64 i + lane = (pixel c / 16, slot c % 16), which holds the
    // pixel's channel chunk slot ^ swz(py, px) (the swizzle goes on the SOURCE address: a DMA writes lane-contiguous bytes);
    // out-of-image halo pixels are out-of-range offsets, written as zeros
    constexpr int WPIECES = WCHUNKS / 64;            // 45
    constexpr int WDS = (WPIECES + 7) / 8;           // 6 per wave
    unsigned goff[WDS];
#pragma unroll
    for (int s = 0; s < WDS; ++s) {
        const int c = 64 * (wv + 8 * s) + lane;
        const int px_ = c >> 4, slot = c & 15;
        const int py = px_ / WPW, pxx = px_ - py * WPW;
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + pxx;
        const bool ok = px_ < WPH * WPW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        goff[s] = ok ? (img_off + (unsigned)(gy * p.W + gx) * (unsigned)p.ld_in * 4u + (unsigned)(slot ^ wino_swz(py, pxx)) * 16u) : kOobW;
    }
    const int rot = p.groups > 1 ? (int)(blockIdx.x % (unsigned)p.groups) : 0;
    auto phys = [&](int g) -> int { const int x = g + rot; return x >= p.groups ? x - p.groups : x; };
    const i32x4 desc = {(int)(unsigned)(size_t)p.in, (int)(((size_t)p.in >> 32) & 0xFFFFu), (int)p.x_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ldsw;
    auto dma_piece = [&](int s, int buf, int g) {
        if (wv + 8 * s < WPIECES) {                  // wave-uniform
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * WPATCH) + 1024u * (unsigned)(wv + 8 * s));
            // the group's 256-byte channel slice enters as the SCALAR offset (not part of the range check: a halo lane's
            // out-of-range per-lane offset stays out of range and its LDS bytes are written as zeros)
            const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)phys(g) * 256u);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(goff[s]), "s"(dst), "s"(desc), "s"(soff) : "memory");
        }
    };

    // this lane's Winograd tile and the 2 x 4 window pixels row i of B^T d needs:  r_b = d[a1][b] + s2 d[a2][b]
    //   i = 0: d0 - d2      i = 1: d1 + d2      i = 2: d2 - d1      i = 3: d1 - d3
    const int tyw = li >> 3, txw = li & 7;
    const int a1 = w == 0 ? 0 : (w == 2 ? 2 : 1), a2 = w == 0 ? 2 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
    const float s2 = w == 1 ? 1.f : -1.f;
    int pbase[2][4], pswz[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int py = 2 * tyw + (r == 0 ? a1 : a2), px = 2 * txw + b;
            pbase[r][b] = (py * WPW + px) * 256;
            pswz[r][b] = wino_swz(py, px);
        }
    // weights: fragment (position 4 w + j, nt) of k-chunk jc = 8 g + jj
    // (through a buffer descriptor: the lane part of the address is ONE constant VGPR, the fragment index a scalar offset -- no
    // 64-bit VALU address arithmetic in the loop)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(p.wf), 0, 0xFFFFFFFFu, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned pos_bytes = (unsigned)p.pos_stride * 16u;
    auto frag = [&](int j, int jc, int blk) -> float4 {
        const unsigned so = (unsigned)(4 * w + j) * pos_bytes + (unsigned)((nth + 2 * (blk0 + blk)) * p.nj + jc) * 1024u;
        const i32x4 x = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)lane16, (int)so, 0);
        return make_float4(__int_as_float(x.x), __int_as_float(x.y), __int_as_float(x.z), __int_as_float(x.w));
    };

#pragma unroll
    for (int s = 0; s < WDS; ++s) dma_piece(s, 0, 0);
    // weight ring of 4 steps indexed by jj % 4 (8 % 4 == 0: static indices), requested WPF steps ahead
    constexpr int WPF = 2;
    float4 wr[4][4];
#pragma unroll
    for (int s0 = 0; s0 < WPF; ++s0)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[s0][j] = frag(j, 8 * phys(0) + s0, 0);
    WPROBE(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WPROBE(2);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    WPROBE(3);

    const int G = p.groups;
    int pbuf = 0;
    // Software pipeline over the 8-channel steps of a group: the window reads and the B^T d B transform of step jj + 1 are issued
    // BETWEEN the four 8-MFMA blocks of step jj (one wave per SIMD: placed in front of its own MFMAs, the ~250-cycle LDS round trip
    // and the ~80 VALU of the transform would sit exposed in front of every 2048-cycle MFMA block).  Only a group's first step
    // (new patch buffer) pays them in the open.
    auto load_d = [&](const unsigned char* pb, int jj, float4 (&d)[2][4]) {
        const int ch = 2 * jj + lh;                                  // this lane's 16-B chunk: channels 8 jj + 4 lh .. + 3
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int b = 0; b < 4; ++b) d[r][b] = *reinterpret_cast<const float4*>(pb + pbase[r][b] + ((ch ^ pswz[r][b]) << 4));
    };
    // the transform in PACKED fp32 (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth per issue slot): every VALU instruction of a SIMD takes
    // ~4 cycles from its fp32 matrix pipe whichever wave issues it (tools/ubench/mfma_valu.hip), so the 32 scalar operations of a
    // step were 12 % of its 1024 MFMA cycles; written on 2-vectors because the library is built without the SLP vectoriser
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 s2v = {s2, s2};
    auto lo = [](const float4& a) -> f32x2 { return f32x2{a.x, a.y}; };
    auto hi = [](const float4& a) -> f32x2 { return f32x2{a.z, a.w}; };
    auto join = [](f32x2 a, f32x2 b) -> float4 { return make_float4(a.x, a.y, b.x, b.y); };
    auto rows = [&](const float4 (&d)[2][4], float4 (&r)[4]) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
            r[b] = join(__builtin_elementwise_fma(s2v, lo(d[1][b]), lo(d[0][b])), __builtin_elementwise_fma(s2v, hi(d[1][b]), hi(d[0][b])));
    };
    auto cols = [&](const float4 (&r)[4], float4 (&v)[4]) {
        v[0] = join(lo(r[0]) - lo(r[2]), hi(r[0]) - hi(r[2]));
        v[1] = join(lo(r[1]) + lo(r[2]), hi(r[1]) + hi(r[2]));
        v[2] = join(lo(r[2]) - lo(r[1]), hi(r[2]) - hi(r[1]));
        v[3] = join(lo(r[1]) - lo(r[3]), hi(r[1]) - hi(r[3]));
    };
    float4 v[2][4];                                                  // transformed inputs of the current / next step
    const int n_blk = TABLE ? p.n_blk : 1;
#pragma unroll 1
    for (int blk = 0; blk < n_blk; ++blk) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const bool more = g + 1 < G;                                 // another input group of this block follows
        const bool more_w = more || blk + 1 < n_blk;                 // ... or another output block: its first weights are requested ahead too
        const unsigned char* pb = ldsw + pbuf * WPATCH;
        {
            float4 d[2][4], r[4];
            load_d(pb, 0, d);
            rows(d, r);
            cols(r, v[0]);
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            if (more && jj < WDS) dma_piece(jj, pbuf ^ 1, g + 1);      // next group's patch, one piece per step, BEFORE this step's weight
                                                                         // requests (vmcnt retires in order: they give it two steps to land)
            {   // weights WPF steps ahead (a step is 16 MFMAs of this wave = 2048 cycles of the shared pipe; every workgroup of the
                // launch walks the same fragments, so the L2 round trip under load is longer than one step)
                int jc = 8 * phys(g) + jj + WPF, nb = blk;
                bool have = true;
                if (jj + WPF >= 8) {
                    have = more_w;
                    jc = (more ? 8 * phys(g + 1) : 8 * phys(0)) + jj + WPF - 8;
                    nb = more ? blk : blk + 1;
                }
                if (have) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) wr[(jj + WPF) & 3][j] = frag(j, jc, nb);
                }
            }
            float4 d[2][4], r[4];
            if (jj < 7) load_d(pb, jj + 1, d);
            __builtin_amdgcn_sched_barrier(0);                       // keep the requests at the top of the step (hipcc sinks them otherwise)
            // (hipcc sinks the transform of step jj + 1 -- its only uses are in the next step's basic block -- in front of that step's
            // MFMAs; pinning one float4 of it behind each MFMA pair with empty asm statements was measured: 54.5 -> 55.8 ms per 8 tiles)
            const float4 (&vc)[4] = v[jj & 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[jj & 3][j].x, vc[j].x, acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (jj < 7) rows(d, r);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[jj & 3][j].y, vc[j].y, acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (jj < 7) cols(r, v[(jj + 1) & 1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[jj & 3][j].z, vc[j].z, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[jj & 3][j].w, vc[j].w, acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            WPROBE(4 + 8 * g + jj);
        }
        if (more) {
            // every DMA piece of the next patch is older than the last 2 x 4 weight requests of this wave
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(4 * WPF) : "memory");
            pbuf ^= 1;
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        WPROBE(20 + g);
    }

    // Output transform.  Column half (A^T along j) in registers: t_x = m_0 + m_1 + m_2 (x = 0), m_1 - m_2 - m_3 (x = 1); row half
    // (A^T along i = wave) through LDS in a fixed order: Y_0 = t[0] + t[1] + t[2], Y_1 = t[1] - t[2] - t[3].
    //   red[i][x][nt][q][lane] = float4 of registers 4 q .. 4 q + 3 (channels 32 nt + 8 q + 4 lh .. + 3 of Winograd tile li)
    // Dense layer: the scratch overlays the patch buffers (all groups are done); logit table: behind the resident patch.
    float4* red = reinterpret_cast<float4*>(ldsw + (TABLE ? WPATCH : 0));
    {
        const int nt = nth;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 t0, t1;
            t0.x = acc[0][4 * q] + acc[1][4 * q] + acc[2][4 * q];
            t0.y = acc[0][4 * q + 1] + acc[1][4 * q + 1] + acc[2][4 * q + 1];
            t0.z = acc[0][4 * q + 2] + acc[1][4 * q + 2] + acc[2][4 * q + 2];
            t0.w = acc[0][4 * q + 3] + acc[1][4 * q + 3] + acc[2][4 * q + 3];
            t1.x = acc[1][4 * q] - acc[2][4 * q] - acc[3][4 * q];
            t1.y = acc[1][4 * q + 1] - acc[2][4 * q + 1] - acc[3][4 * q + 1];
            t1.z = acc[1][4 * q + 2] - acc[2][4 * q + 2] - acc[3][4 * q + 2];
            t1.w = acc[1][4 * q + 3] - acc[2][4 * q + 3] - acc[3][4 * q + 3];
            red[(((w * 2 + 0) * 2 + nt) * 4 + q) * 64 + lane] = t0;
            red[(((w * 2 + 1) * 2 + nt) * 4 + q) * 64 + lane] = t1;
        }
    }
    WPROBE(28);
    __syncthreads();
    WPROBE(29);
    float* const xi = TABLE ? p.x + (size_t)img * p.out_y_stride : p.x + (size_t)img * p.H * p.W * p.ldx;
    // this thread's four output units share their channel quad: ONE bias load in front of the loop (inside it, behind the bounds
    // test and next to the stores it may alias, hipcc keeps load -> store order and the L2 round trip is paid per unit)
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (!TABLE) bias4 = *reinterpret_cast<const float4*>(p.bias + 32 * ((t >> 8) & 1) + 8 * ((t >> 6) & 3) + 4 * ((t & 63) >> 5));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int unit = t + 512 * u;                  // (y, x, nt, q, lane): 2 x 2 x 2 x 4 x 64
        const int ul = unit & 63, q = (unit >> 6) & 3, nt = (unit >> 8) & 1, xx = (unit >> 9) & 1, yy = unit >> 10;
        auto rd = [&](int i) -> float4 { return red[(((i * 2 + xx) * 2 + nt) * 4 + q) * 64 + ul]; };
        float4 o;
        if (yy == 0) {
            const float4 a = rd(0), b = rd(1), c = rd(2);
            o = make_float4(a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z, a.w + b.w + c.w);
        } else {
            const float4 a = rd(1), b = rd(2), c = rd(3);
            o = make_float4(a.x - b.x - c.x, a.y - b.y - c.y, a.z - b.z - c.z, a.w - b.w - c.w);
        }
        const int tl = ul & 31;
        const int y = ty0 + 2 * (tl >> 3) + yy, x = tx0 + 2 * (tl & 7) + xx;
        if (y < p.H && x < p.W) {
            const int co = 32 * nt + 8 * q + 4 * (ul >> 5);
            if constexpr (!TABLE) {
                const float4 b = bias4;                 // co is the same for every unit of a thread (512 u only moves the pixel)
                o.x = fmaxf(o.x + b.x, 0.f); o.y = fmaxf(o.y + b.y, 0.f);
                o.z = fmaxf(o.z + b.z, 0.f); o.w = fmaxf(o.w + b.w, 0.f);
            }
            *reinterpret_cast<float4*>(xi + ((size_t)y * p.W + x) * p.ldx + p.col_out + 64 * (blk0 + blk) + co) = o;
        }
    }
    if (TABLE && blk + 1 < n_blk) __syncthreads();      // the scratch is rewritten by the next block's output transform
    }
    WPROBE(30);
}

int dense_wino_tiles(int H, int W) { return ceil_div(H, WTH) * ceil_div(W, WTW); }

constexpr size_t kWinoTableLds = (size_t)WPATCH + 65536;      // the resident patch + the 64-KB output-transform scratch behind it

// dense layer l of a block in Winograd form; frag_wino = 16 fragment arrays of the transformed weights (encoder_hip.py packs them)
int dense_layer_wino_f32(float* X, int ldx, int H, int W, int l, const float* frag_wino, const float* bias, int n_img, hipStream_t s) {
    CIAOSR_CHECK_ARG(X && frag_wino && bias && (ldx & 3) == 0 && aligned16(X) && aligned16(frag_wino) && aligned16(bias));
    const size_t x_bytes = (size_t)n_img * H * W * ldx * 4;
    CIAOSR_CHECK_ARG(n_img >= 1 && n_img <= 65535 && x_bytes < 0xFFFFFF00ull);
    DenseWinoP p;
    p.x = X; p.ldx = ldx; p.x_bytes = (unsigned)x_bytes;
    p.in = X; p.ld_in = ldx; p.n_blk = 1; p.out_y_stride = 0;
    p.H = H; p.W = W; p.tiles_x = ceil_div(W, WTW);
    p.groups = l + 1;
    p.wf = reinterpret_cast<const float4*>(frag_wino);
    p.nj = 64 * (l + 1) / 8;
    p.pos_stride = (long)2 * p.nj * 64;
    p.bias = bias;
    p.col_out = 64 * (l + 1);
    CIAOSR_BIG_LDS(dense_wino_f32_kernel<false>, kWinoLds);
    ProfScope prof("enc_dense_wino", s);
    hipLaunchKernelGGL(dense_wino_f32_kernel<false>, dim3(dense_wino_tiles(H, W), n_img), dim3(512), kWinoLds, s, p);
    return launch_status("dense_wino_f32");
}

// Nine 3x3 convolutions 64 -> 64 n_blk channels without bias: out[(pix * 9 + o) * ldg + n] = sum_{k, c} Pi[o][pix + k][c] w[n][c][k]
// (the logit table of the fused head: Pi_o = F . shift_o(F), head.hip).  Pi: [9][H*W][64]; frag_wino: 16 arrays of the transformed
// [64 n_blk][64] weights in fragment order.
int wino_table_f32(const float* Pi, int H, int W, const float* frag_wino, int n_blk, float* out, int ldg, hipStream_t s) {
#ifdef CIAOSR_PROBE       // developer A/B only: the product library reads no environment
    static const int per_wg = [] { const char* e = getenv("CIAOSR_TABLE_BLK"); const int v = e ? atoi(e) : 4; return v == 1 || v == 2 ? v : 4; }();
#else
    constexpr int per_wg = 4;
#endif
    CIAOSR_CHECK_ARG(Pi && frag_wino && out && n_blk >= 1 && (ldg & 3) == 0 && aligned16(Pi) && aligned16(frag_wino) && aligned16(out));
    const size_t in_bytes = (size_t)9 * H * W * 64 * 4;
    CIAOSR_CHECK_ARG(in_bytes < 0xFFFFFF00ull);
    DenseWinoP p;
    p.x = out; p.ldx = 9 * ldg; p.x_bytes = (unsigned)in_bytes;
    CIAOSR_CHECK_ARG(n_blk % per_wg == 0);
    p.in = Pi; p.ld_in = 64; p.n_blk = per_wg; p.out_y_stride = ldg;
    p.H = H; p.W = W; p.tiles_x = ceil_div(W, WTW);
    p.groups = 1;
    p.wf = reinterpret_cast<const float4*>(frag_wino);
    p.nj = 8;
    p.pos_stride = (long)2 * n_blk * p.nj * 64;
    p.bias = nullptr;
    p.col_out = 0;
    CIAOSR_BIG_LDS(dense_wino_f32_kernel<true>, kWinoTableLds);
    ProfScope prof("head_logit_table_w2", s);   // F(2x2) form: 16 / 36
    hipLaunchKernelGGL(dense_wino_f32_kernel<true>, dim3(dense_wino_tiles(H, W), 9, n_blk / per_wg), dim3(512), kWinoTableLds, s, p);
    return launch_status("wino_table_f32");
}

}  // namespace ciaosr

#ifdef CIAOSR_PROBE
extern "C" int ciaosr_debug_wino_probe_read(unsigned long long* host, int n_words, int next_groups) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ciaosr::g_wprobe), (size_t)n_words * 8) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(ciaosr::g_wprobe_groups), &next_groups, sizeof(int)) == hipSuccess ? 0 : -1;
}
#endif
