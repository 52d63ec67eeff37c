// 16-bit-MFMA dense-block convolution (precision modes "bf16" / "f16" of the RDN trunk, big tiles only; compiled once per
// element type, h16_util.h -- "bf16" below = the 16-bit element type of the build).
//
// One 3x3 dense layer l of a residual dense block (mmedit RDB.layers[l].conv over cat(x, d_0 .. d_{l-1}),
// called from ciaosr_net.py:330-337) in GATHER form: K = 9 * 64 (l+1), N = 64 output channels,
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Inputs are the bf16 copy Xb [HW][64*9] of the block's
// feature buffer; the output relu(conv + b) is written as fp32 (for the fp32 LFF / residual path) and as bf16
// (for the following dense layers).
//
// The fp32 convolution of conv_f32.hip is bound by its operand stream (one 64-channel K-stage of A and B is
// fetched per 64x64x64 MACs), not by the MFMA, so a 16x faster MFMA alone would buy nothing.  Here the traffic
// per MAC is cut instead:
//   * workgroup = 12x12 output pixels x all 64 output channels; a 192x192 tile is exactly 256 workgroups,
//     one per CU;
//   * the 14x14-pixel halo patch of one 64-channel input group (bf16, 25 KB) is loaded into LDS ONCE and
//     serves all 9 taps (the next group's patch is prefetched into registers meanwhile, two LDS buffers);
//     round 2: the patch loads leave in one burst after the tap-3 weights (vmcnt retires in order, so a patch load
//     issued every tap made every weight wait also a wait for HBM), the LDS reads of tap t+1 are issued before the
//     MFMAs of tap t (hipcc had placed them directly in front of their use: ~150 exposed cycles per 320-cycle tap),
//     and the patch row pitch is padded to a bank-conflict-free 2240 B;
//   * the 4 waves split K (wave w owns channels 16w..16w+15 of every group): each wave accumulates the whole
//     160(144 used) x 64 tile -- 5x2 MFMA tiles, 160 accumulator registers -- so one 16-B LDS read per pixel
//     tile feeds two MFMAs and one 1-KB weight fragment read from L2 (pre-packed, coalesced, no LDS) feeds five;
//     LDS traffic is ~75 B/clk/CU at the full MFMA rate, under the 128 B/clk limit;
//   * the K-slices are summed through LDS once per layer (fixed order: deterministic) in the epilogue.
// Swapped operands (weights = A, activations = B): a lane owns one pixel and 4x4 consecutive channels.

#include <cstdlib>

#include "h16_util.h"
#include "ops.h"

namespace ciaosr {
namespace CIAOSR_H16_NS {

constexpr bool kF16 = CIAOSR_F16 != 0;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int DT = 12;                       // output tile edge (pixels)
constexpr int DP = DT + 2;                   // patch edge with the 1-pixel halo
constexpr int DPS = 144;                     // bytes per patch pixel: 64 bf16 + 16 B pad (conflict-free ds_read_b128)
constexpr int DROW = 2240;                   // bytes per patch row: 14 x 144 + 224 B pad.  With the 144-B pixel stride a ds_read_b128 of
                                             // 16 lanes is conflict-free iff the lanes' (9 x + 12 y) mod 16 differ; a pitch = 192 mod 256
                                             // gives every 16-lane group of every pixel tile distinct bank quads (round 1's 2016-B
                                             // pitch: 34 two-way conflicts per 160 lane reads, SQ_LDS_BANK_CONFLICT 32 % of LDS cycles)
constexpr int DPATCH = DP * DROW;            // 31 360 B per buffer
constexpr int DCHUNKS = DP * DP * 8;         // 16-byte chunks of a patch
constexpr int DLOADS = (DCHUNKS + 255) / 256;
constexpr int DMT = 5;                       // 32-pixel MFMA tiles per workgroup (160 rows, 144 used)
constexpr size_t kDenseLds = 81920;          // max(2 patches, K-slice reduction scratch 4 x 5 x 4 x 64 x 16 B)
constexpr unsigned kOobD = 0xFFFFFFF0u;

struct DenseH16P {
    const unsigned short* xb; int ldxb;      // bf16 feature buffer [HW][ldxb], 64-channel groups
    unsigned xb_bytes;
    int H, W, tiles_x;
    int n_img;                               // images in the buffer (back to back); one workgroup walks its tile of every image
    int groups;                              // input groups of this layer (l + 1)
    const uint4* wf; int nks;                // ciaosr_pack_fragments_bf16 of the conv weight [64][9*cin]: [2][nks][64 lanes]
    const uint4* wf_lo;                      // ciaosr_pack_fragments_bf16_lo of the same matrix (hi + lo weight pair), or null
    const float* bias;                       // [64]
    float* x; int ldx;                       // fp32 feature buffer (written at column col_out), or null
    unsigned short* xb_out;                  // == xb (written at column col_out)
    int col_out;
};

#ifndef CIAOSR_DENSE_PF
#define CIAOSR_DENSE_PF 4
#endif
constexpr int DPF = CIAOSR_DENSE_PF;          // weight fragments are requested this many taps ahead (1..8)

// LO: second MFMA per product with the low halves of the weight pairs (bf16 default); compiled out otherwise so that its
// registers go to a deeper weight pipeline
template <bool LO>
__global__ __launch_bounds__(256) void dense_h16_kernel(DenseH16P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
    const int ty0 = (blockIdx.x / p.tiles_x) * DT, tx0 = (blockIdx.x % p.tiles_x) * DT;
    // The workgroup walks its 12x12 tile of EVERY image of the batch (same weights, own rows) as one software pipeline: the first
    // weights and the halo patch of image i + 1 are requested during the last input group of image i, so only the first image pays the
    // cold start.  Per image the work, its order and hence the result are those of a single-image launch.
    const unsigned img_bytes = (unsigned)((size_t)p.H * p.W * p.ldxb * 2);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.xb), 0, p.xb_bytes, 0x00020000);

    // patch staging: thread -> 16-byte chunks t + 256 s  (pixel = chunk / 8, 8 chunks = 64 channels)
    unsigned goff[DLOADS];
    int loff[DLOADS];
#pragma unroll
    for (int s = 0; s < DLOADS; ++s) {
        const int c = t + 256 * s;
        const int px = c >> 3, part = c & 7;
        const int py = px / DP, pxx = px - py * DP;
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + pxx;
        const bool ok = c < DCHUNKS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        goff[s] = ok ? ((unsigned)(gy * p.W + gx) * (unsigned)p.ldxb * 2u + (unsigned)part * 16u) : kOobD;
        loff[s] = c < DCHUNKS ? py * DROW + pxx * DPS + part * 16 : -1;
    }
    // Every workgroup of a layer walks the same weights: started in lockstep, all 256 would pull the same L2 lines at the
    // same time (hot channels).  Workgroup b therefore walks the input groups in the rotated order (g + b) mod G -- a
    // different summation order per workgroup position (fp32 accumulation; deterministic), the same MACs.
    const int rot = p.groups > 1 ? (int)(blockIdx.x % (unsigned)p.groups) : 0;
    auto phys = [&](int g) -> int { const int x = g + rot; return x >= p.groups ? x - p.groups : x; };
    i32x4 P[DLOADS];
    auto load_chunk = [&](int s, int g, int img) {
        P[s] = __builtin_amdgcn_raw_buffer_load_b128(
            rs, goff[s] == kOobD ? (int)kOobD : (int)(goff[s] + (unsigned)phys(g) * 128u + (unsigned)img * img_bytes), 0, 0);
    };
    auto store_patch = [&](int buf) {
#pragma unroll
        for (int s = 0; s < DLOADS; ++s)
            if (loff[s] >= 0) *reinterpret_cast<i32x4*>(lds + buf * DPATCH + loff[s]) = P[s];
    };

    // B operand (activations): pixel of lane li in each of the 5 pixel tiles, this wave's 16-channel K slice
    int poff[DMT];
#pragma unroll
    for (int r = 0; r < DMT; ++r) {
        int idx = 32 * r + li;
        idx = idx < DT * DT ? idx : DT * DT - 1;
        const int y = idx / DT, x = idx - y * DT;
        poff[r] = (y + 1) * DROW + (x + 1) * DPS + (16 * w + 8 * lh) * 2;
    }
    const uint4* wl = p.wf + lane;
    const int kpt = 4 * p.groups;            // k16-steps per tap (cin / 16)
    auto frag = [&](int nt, int g, int tap) -> uint4 { return wl[(size_t)(nt * p.nks + tap * kpt + 4 * phys(g) + w) * 64]; };
    constexpr bool has_lo = LO;
    const uint4* wll = (has_lo ? p.wf_lo : p.wf) + lane;
    auto frag_lo = [&](int nt, int g, int tap) -> uint4 { return wll[(size_t)(nt * p.nks + tap * kpt + 4 * phys(g) + w) * 64]; };

    f32x16 acc[2][DMT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < DMT; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][r][e] = 0.f;
    };
    zero_acc();

    // prologue: the first DPF weight stages go out BEFORE the patch (vmcnt retires in order: a later wait on the
    // weights would otherwise also wait for the patch)
    uint4 wq[9][2], wlq[9][2];               // weight fragments (hi, lo) by tap; DPF + 1 of them are live at a time
#pragma unroll
    for (int tp = 0; tp < DPF; ++tp) {
        wq[tp][0] = frag(0, 0, tp); wq[tp][1] = frag(1, 0, tp);
        if (has_lo) { wlq[tp][0] = frag_lo(0, 0, tp); wlq[tp][1] = frag_lo(1, 0, tp); }
    }
#pragma unroll
    for (int s = 0; s < DLOADS; ++s) load_chunk(s, 0, 0);
    store_patch(0);
    __syncthreads();

    const int G = p.groups;
    uint4 b[2][DMT];                         // activation fragments of the current tap and the next one
    int pbuf = 0;                            // LDS patch buffer of the current (image, group)
#pragma unroll 1
    for (int img = 0; img < p.n_img; ++img) {
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const bool last_g = g + 1 == G;
        const bool more = !last_g || img + 1 < p.n_img;          // another (image, group) follows: prefetch it
        const int ng_next = last_g ? 0 : g + 1, nimg_next = last_g ? img + 1 : img;
        const unsigned char* pb = lds + pbuf * DPATCH;
#pragma unroll
        for (int r = 0; r < DMT; ++r)
            b[0][r] = *reinterpret_cast<const uint4*>(pb + poff[r] - DROW - DPS);      // tap 0
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            {   // weights DPF taps ahead
                int ng = g, ntap = tap + DPF;
                bool have = true;
                if (ntap >= 9) { ntap -= 9; ng = ng_next; have = more; }      // the next group -- of this image or the first of the next
                if (have) {
                    wq[(tap + DPF) % 9][0] = frag(0, ng, ntap); wq[(tap + DPF) % 9][1] = frag(1, ng, ntap);
                    if (has_lo) { wlq[(tap + DPF) % 9][0] = frag_lo(0, ng, ntap); wlq[(tap + DPF) % 9][1] = frag_lo(1, ng, ntap); }
                }
            }
            // next group's whole patch right after the tap-3 weights left: the first wait that covers it is tap 4's
            if (more && tap == 1) {
#pragma unroll
                for (int s = 0; s < DLOADS; ++s) load_chunk(s, ng_next, nimg_next);
            }
            if (tap + 1 < 9) {               // LDS reads one tap ahead: their latency runs under this tap's 10 MFMAs
                const int toff = (((tap + 1) / 3 - 1) * DROW) + (((tap + 1) % 3 - 1) * DPS);
#pragma unroll
                for (int r = 0; r < DMT; ++r)
                    b[(tap + 1) & 1][r] = *reinterpret_cast<const uint4*>(pb + poff[r] + toff);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const uint4 a = wq[tap][nt];
#pragma unroll
                for (int r = 0; r < DMT; ++r) acc[nt][r] = mfma_h16<kF16>(a, b[tap & 1][r], acc[nt][r]);
            }
            if (has_lo) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const uint4 a = wlq[tap][nt];
#pragma unroll
                    for (int r = 0; r < DMT; ++r) acc[nt][r] = mfma_h16<kF16>(a, b[tap & 1][r], acc[nt][r]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the ring is indexed by tap: stages 0 .. DPF-1 now hold the first taps of the next group
        if (!last_g) {
            store_patch(pbuf ^ 1);
            pbuf ^= 1;
        }
        __syncthreads();
    }

    // K-slice reduction + epilogue of this image, one 32-channel half at a time through LDS (the scratch overlays both patch buffers:
    // the next image's patch stays in registers until it is done):
    // red[w][r][q][lane] = float4 of accumulator registers 4q..4q+3 (= channels 8q + 4lh .. +3 of pixel li of tile r)
    const size_t ipix = (size_t)img * p.H * p.W;
    float4* red = reinterpret_cast<float4*>(lds);
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int r = 0; r < DMT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                red[((w * DMT + r) * 4 + q) * 64 + lane] =
                    make_float4(acc[nt][r][4 * q], acc[nt][r][4 * q + 1], acc[nt][r][4 * q + 2], acc[nt][r][4 * q + 3]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < DMT; ++u) {
            // unit = (pixel, 4-channel group): the eight lanes of a pixel write the 128 B (fp32) / 64 B (16-bit) of its 32 channels of this
            // pass -- whole lines, 8 pixels per store instruction instead of 32 pixels x 32 B (see the dma kernel below)
            const int unit = t + 256 * u;
            const int pidx = unit >> 3, oct = unit & 7, q = oct >> 1, r = pidx >> 5, ul = ((oct & 1) << 5) | (pidx & 31);
            float4 v = red[((0 * DMT + r) * 4 + q) * 64 + ul];
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) {
                const float4 o = red[((ww * DMT + r) * 4 + q) * 64 + ul];
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            const int idx = 32 * r + (ul & 31);
            const int y = ty0 + idx / DT, x = tx0 + idx % DT;
            if (idx < DT * DT && y < p.H && x < p.W) {
                const int co = 32 * nt + 8 * q + 4 * (ul >> 5);
                const float4 b = *reinterpret_cast<const float4*>(p.bias + co);
                v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f);
                v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
                const size_t pix = ipix + (size_t)y * p.W + x;
                if (p.x) *reinterpret_cast<float4*>(p.x + pix * p.ldx + p.col_out + co) = v;
                *reinterpret_cast<uint2*>(p.xb_out + pix * p.ldxb + p.col_out + co) =
                    pack_h16x4<kF16>(v.x, v.y, v.z, v.w);
            }
        }
        __syncthreads();
    }
    if (img + 1 < p.n_img) {                 // next image: its first patch (requested during the last group) goes to LDS now
        zero_acc();
        pbuf = 0;
        store_patch(0);
        __syncthreads();
    }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 3: the same layer with TWO workgroups per CU (one 16-bit weight per product: f16, or bf16-single).
//
// The kernel above holds one workgroup per CU (142 VGPR + 160 accumulators): its per-workgroup fixed part -- first weights and halo
// patch from L2 / HBM, the K-slice reduction through LDS, the stores (8.6 us of a 19 us average launch) -- and every exposed
// latency inside the group loop (SQ counters: matrix pipe 21 % busy, 39 % of the wave cycles parked in s_waitcnt / barriers)
// stall the matrix pipe directly.  Here the registers beside the 160 accumulators are cut to < 96, so that a second, independent
// workgroup shares the CU and fills those holes:
//   * the halo patch goes from memory STRAIGHT to LDS (`buffer_load_dwordx4 ... lds`, 25 x 1 KB pieces): no 28 staging + 14
//     address registers.  A DMA writes lane-contiguous bytes, so the patch is an UNPADDED [14][14] array of 128-B pixels whose eight
//     16-B channel chunks are XOR-swizzled with g(py, px) = ((px >> 1) - 2 py) & 7 -- applied to the SOURCE address of the DMA and
//     to the ds_read_b128 address alike.  Brute force over every (tap, pixel tile, K slice, 16-lane group of ds_read_b128): the 16
//     lanes of a group always hit 16 distinct bank quads (the padded 2240-B pitch of the kernel above is not DMA-compatible);
//     out-of-image halo pixels are out-of-range buffer offsets, which the DMA writes as zeros;
//   * activation fragments single-buffered: tile r of the next tap is read right behind the two MFMAs that use tile r of this
//     one; weight ring 3 taps (requested 2 ahead; the partner workgroup covers the rest of the L2 latency);
//   * the K-slice reduction runs in four 40-KB passes in the LDS the NEXT patch does not occupy, so the next image's first patch
//     can land during the epilogue (66 560 B of LDS per workgroup: two per CU);
//   * grid.y = image slices: with a batch of tiles every CU gets two workgroups that walk different images.
// Per image the MACs and their order are those of the kernel above (and of a single-image launch): bitwise the same result.
constexpr int D2_PATCH = 25 * 1024;              // 1568 chunks of 16 B, rounded up to whole 1-KB DMA pieces
constexpr int D2_FREE = 15360;
constexpr int D2_P1 = D2_PATCH + D2_FREE;        // [P0][free][P1]
constexpr size_t kDense2Lds = 2 * D2_PATCH + D2_FREE;      // 66 560 B
constexpr int D2_RED = 4 * DMT * 2 * 64 * 16;    // one reduction pass: 4 waves x 5 tiles x 2 register quads x 64 lanes x 16 B = 40 960
static_assert(D2_RED <= D2_PATCH + D2_FREE, "reduction scratch must fit beside the next patch");

__global__ __launch_bounds__(256, 2) void dense_h16_dma_kernel(DenseH16P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, lh = lane >> 5;
    const int ty0 = (blockIdx.x / p.tiles_x) * DT, tx0 = (blockIdx.x % p.tiles_x) * DT;
    // images of this workgroup: slice blockIdx.y of gridDim.y
    const int per = (p.n_img + (int)gridDim.y - 1) / (int)gridDim.y;
    const int img0 = (int)blockIdx.y * per, img1 = img0 + per < p.n_img ? img0 + per : p.n_img;
    if (img0 >= img1) return;
    const unsigned img_bytes = (unsigned)((size_t)p.H * p.W * p.ldxb * 2);
    const i32x4 desc = {(int)(unsigned)(size_t)p.xb, (int)(((size_t)p.xb >> 32) & 0xFFFFu), (int)p.xb_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

    // DMA pieces of this wave: piece i = w + 4 s (s < 7, i < 25) fills LDS bytes [1024 i, 1024 i + 1024) of a patch buffer; lane ->
    // LDS chunk c = 64 i + lane = (pixel c / 8, slot c % 8), which holds the pixel's channel chunk j = slot ^ g(py, px)
    constexpr int D2S = 7;
    unsigned goff[D2S];
#pragma unroll
    for (int s = 0; s < D2S; ++s) {
        const int c = 64 * (w + 4 * s) + lane;
        const int px_ = c >> 3, slot = c & 7;
        const int py = px_ / DP, pxx = px_ - py * DP;
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + pxx;
        const bool ok = px_ < DP * DP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const int j = slot ^ (((pxx >> 1) - 2 * py) & 7);
        goff[s] = ok ? ((unsigned)(gy * p.W + gx) * (unsigned)p.ldxb * 2u + (unsigned)j * 16u) : kOobD;
    }
    const int rot = p.groups > 1 ? (int)(blockIdx.x % (unsigned)p.groups) : 0;
    auto phys = [&](int g) -> int { const int x = g + rot; return x >= p.groups ? x - p.groups : x; };
    auto dma_patch = [&](int buf, int g, int img) {
        const unsigned add = (unsigned)phys(g) * 128u + (unsigned)img * img_bytes;
#pragma unroll
        for (int s = 0; s < D2S; ++s) {
            if (w + 4 * s < 25) {                     // wave-uniform
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf ? D2_P1 : 0) + 1024u * (unsigned)(w + 4 * s));
                const unsigned voff = goff[s] == kOobD ? kOobD : goff[s] + add;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(dst), "s"(desc) : "memory");
            }
        }
    };

    // B operand (activations): centre pixel of lane li in each of the 5 pixel tiles.  pbase = byte offset of the pixel, hpk = the three
    // swizzle keys (px >> 1) - 2 py (+ the dx = -1 / 0 / +1 corrections), 4 bits each
    int pbase[DMT], hpk[DMT];
#pragma unroll
    for (int r = 0; r < DMT; ++r) {
        int idx = 32 * r + li;
        idx = idx < DT * DT ? idx : DT * DT - 1;
        const int y = idx / DT, x = idx - y * DT;
        const int py = y + 1, pxx = x + 1;
        pbase[r] = (py * DP + pxx) * 128;
        const int h = (pxx >> 1) - 2 * py, par = pxx & 1;
        hpk[r] = ((h + par - 1) & 7) | ((h & 7) << 4) | (((h + par) & 7) << 8);
    }
    const int jch = 2 * w + lh;                  // this lane's 16-B channel chunk of a pixel: the wave's K slice, this half's 8 channels
    auto b_addr = [&](int r, int tap) -> int {   // tap = 3 (dy + 1) + (dx + 1), compile-time
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int key = ((hpk[r] >> (4 * (dx + 1))) - 2 * dy) & 7;
        return pbase[r] + (dy * DP + dx) * 128 + ((jch ^ key) << 4);
    };
    const uint4* wl = p.wf + lane;
    const int kpt = 4 * p.groups;
    auto frag = [&](int nt, int g, int tap) -> uint4 { return wl[(size_t)(nt * p.nks + tap * kpt + 4 * phys(g) + w) * 64]; };

    f32x16 acc[2][DMT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < DMT; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][r][e] = 0.f;
    };
    zero_acc();

    constexpr int PF = 2, RING = 3;              // weights requested PF taps ahead, ring indexed by tap % RING (9 % 3 == 0)
    uint4 wq[RING][2];
#pragma unroll
    for (int tp = 0; tp < PF; ++tp) { wq[tp][0] = frag(0, 0, tp); wq[tp][1] = frag(1, 0, tp); }
    dma_patch(0, 0, img0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    const int G = p.groups;
    uint4 b[DMT];
    int pbuf = 0;
#pragma unroll 1
    for (int img = img0; img < img1; ++img) {
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const bool last_g = g + 1 == G;
        const bool more = !last_g || img + 1 < img1;
        const int ng_next = last_g ? 0 : g + 1, nimg_next = last_g ? img + 1 : img;
        const unsigned char* pb = lds + (pbuf ? D2_P1 : 0);
        // keep the 45 (tile, tap) read addresses out of registers: without this hipcc hoists all of them out of the group loop
#pragma unroll
        for (int r = 0; r < DMT; ++r) asm volatile("" : "+v"(pbase[r]), "+v"(hpk[r]));
#pragma unroll
        for (int r = 0; r < DMT; ++r) b[r] = *reinterpret_cast<const uint4*>(pb + b_addr(r, 0));
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            {   // weights PF taps ahead
                int ng = g, ntap = tap + PF;
                bool have = true;
                if (ntap >= 9) { ntap -= 9; ng = ng_next; have = more; }
                if (have) { wq[(tap + PF) % RING][0] = frag(0, ng, ntap); wq[(tap + PF) % RING][1] = frag(1, ng, ntap); }
            }
            if (more && tap == 1) dma_patch(pbuf ^ 1, ng_next, nimg_next);      // the other buffer: last read before the previous barrier
#pragma unroll
            for (int r = 0; r < DMT; ++r) {
                acc[0][r] = mfma_h16<kF16>(wq[tap % RING][0], b[r], acc[0][r]);
                acc[1][r] = mfma_h16<kF16>(wq[tap % RING][1], b[r], acc[1][r]);
                if (tap + 1 < 9) b[r] = *reinterpret_cast<const uint4*>(pb + b_addr(r, tap + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!last_g) {
            // the next patch (DMAs of tap 1) has landed once at most the 2 PF weight loads issued after it are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * PF) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            pbuf ^= 1;
        }
    }
    // K-slice reduction + epilogue of this image in four passes (output half nt, register-quad pair qh) through the LDS that the NEXT
    // patch (buffer pbuf ^ 1, possibly still landing) does not occupy.  red[w][r][q2][lane] = accumulator registers 4 (2 qh + q2) .. + 3
    {
        const size_t ipix = (size_t)img * p.H * p.W;
        float4* red = reinterpret_cast<float4*>(lds + (pbuf ? (int)kDense2Lds - D2_RED : 0));
        __syncthreads();                          // every wave is done reading the current patch
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int nt = pass >> 1, qh = pass & 1;
#pragma unroll
            for (int r = 0; r < DMT; ++r)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int q = 2 * qh + q2;
                    red[((w * DMT + r) * 2 + q2) * 64 + lane] =
                        make_float4(acc[nt][r][4 * q], acc[nt][r][4 * q + 1], acc[nt][r][4 * q + 2], acc[nt][r][4 * q + 3]);
                }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                // unit = (pixel idx, channel quad): the four lanes of a pixel write 64 B (fp32) / 32 B (16-bit) of ITS row, so a store
                // instruction covers 16 pixels instead of 32 (the lane = pixel order of the accumulators put 32 pixels x 32 B -- 32
                // quarter-used lines -- into every instruction: the vector L1's tag rate)
                const int unit = t + 256 * u;             // 640 units
                if (unit < DMT * 2 * 64) {
                    const int pidx = unit >> 2, quad = unit & 3, q2 = quad >> 1, r = pidx >> 5, ul = ((quad & 1) << 5) | (pidx & 31);
                    float4 v = red[((0 * DMT + r) * 2 + q2) * 64 + ul];
#pragma unroll
                    for (int ww = 1; ww < 4; ++ww) {
                        const float4 o = red[((ww * DMT + r) * 2 + q2) * 64 + ul];
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    const int idx = 32 * r + (ul & 31);
                    const int y = ty0 + idx / DT, x = tx0 + idx % DT;
                    if (idx < DT * DT && y < p.H && x < p.W) {
                        const int co = 32 * nt + 8 * (2 * qh + q2) + 4 * (ul >> 5);
                        const float4 bb = *reinterpret_cast<const float4*>(p.bias + co);
                        v.x = fmaxf(v.x + bb.x, 0.f); v.y = fmaxf(v.y + bb.y, 0.f);
                        v.z = fmaxf(v.z + bb.z, 0.f); v.w = fmaxf(v.w + bb.w, 0.f);
                        const size_t pix = ipix + (size_t)y * p.W + x;
                        if (p.x) *reinterpret_cast<float4*>(p.x + pix * p.ldx + p.col_out + co) = v;
                        *reinterpret_cast<uint2*>(p.xb_out + pix * p.ldxb + p.col_out + co) = pack_h16x4<kF16>(v.x, v.y, v.z, v.w);
                    }
                }
            }
            __syncthreads();
        }
    }
    if (img + 1 < img1) {                         // next image: its first patch (requested during the last group) is in buffer pbuf ^ 1
        zero_acc();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        pbuf ^= 1;
    }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: the layer re-cut around the L2 -> CU DELIVERY limit (dense_h16_wide_kernel).
//
// Both kernels above stream every weight fragment of the layer from L2 into EVERY 12x12-pixel workgroup: per 64-channel input group a
// workgroup moves 72 KB of weights + 25 KB of halo patch for 2592 CU-cycles of MFMA issue = 37 B/clk per CU, against the 21 B/clk per CU that
// L2 delivers to fragments every CU streams (tools/ubench/l2_stream.hip): an operand-delivery ceiling of ~57 % before any latency, and the
// K-slice reduction through LDS (164 KB written and read back per workgroup and layer) on top.  Here
//   * a workgroup = 512 threads = ONE per CU owns a 16 x 32-pixel tile (16 MFMA pixel tiles: one per tile row) x all 64 output channels:
//     3.6x the pixels per weight byte, halo factor 1.20 instead of 1.36 -> 16 B/clk per CU at the full MFMA rate;
//   * the waves split PIXELS, not K: wave w owns tile rows 2w, 2w + 1 and both output-channel halves (4 accumulator tiles, 64 registers);
//     every product of an output is summed inside one accumulator -- no K-slice reduction, no second pass through LDS;
//   * weights AND halo patch of a STAGE -- 32 input channels (two k16 steps; 16 channels when every weight is a hi + lo pair) -- come by
//     LDS-DMA (`buffer_load_dwordx4 ... lds`, 1-KB pieces: a pre-packed weight fragment IS one piece) into one of two stage buffers while
//     the previous stage computes; all eight waves read the same weight fragments from LDS (ds_read_b128 of lane-contiguous 1-KB blocks:
//     conflict-free), activations from the patch, whose 16-B channel chunks are XOR-swizzled with the pixel index on the SOURCE side of the
//     DMA so that every 16-lane group of a ds_read_b128 touches 16 distinct bank quads (brute-forced below in wide_swizzle_ok);
//     LDS reads: 2 weight + 2 activation fragments per 4 MFMAs = 128 B/clk per CU at the full rate (half the LDS peak);
//   * one barrier per stage; the DMA pieces of stage n + 1 (<= 10 per wave) are issued one per tap between the MFMAs of stage n;
//   * PERSISTENT workgroups walk (image, tile) items b, b + grid, ...: the next item's first stage lands during the current item's last
//     one, and the batch fills the chip evenly (7 images x 72 tiles of a 192x192 map = 504 items on 256 CUs: 98 %);
//   * epilogue per item: bias + ReLU in the accumulator layout, then a per-wave 4-KB LDS transpose so that the stores leave as whole
//     128-B lines (16-bit copy always; fp32 copy when the caller keeps one).
// Per image the MACs and their order depend on the tile position only (stage order rotated by the tile index, like above): a tile's
// result does not depend on the batch it is computed in.  Summation order differs from the 12x12 kernels (no K split): same products.
constexpr int WTW = 32;                         // output tile width (pixels): one MFMA pixel tile per tile row
constexpr int WPW = WTW + 2;                     // with the 1-pixel halo
// R = tile rows per wave: 2 (16 x 32-pixel tiles, the throughput shape) or 1 (8 x 32: twice the workgroups for launches that cannot fill the
// chip with the big tiles -- a single image is 72 tiles of 16 x 32 on 256 CUs).  The accumulation order of an output -- stages in the order
// rotated by the index of its 16 x 32 PARENT tile, taps, k-steps -- is the same in both shapes: the results are bitwise equal, so the shape
// may be chosen by the size of the launch without a tile's result depending on its batch.
template <int KS, int R> struct WideGeom {       // KS = k16 steps per stage: 2 (one 16-bit weight per product) or 1 (hi + lo pairs)
    static constexpr int WTH = 8 * R;                                   // output tile height: one row per (wave, r)
    static constexpr int WPH = WTH + 2;
    static constexpr int WNPIX = WPH * WPW;                             // 612 / 340
    static constexpr int NW = 2 / KS;                                   // weight sets
    static constexpr int CPP = 2 * KS;                                  // 16-B chunks per patch pixel and stage
    static constexpr int PIX_BYTES = 16 * CPP;
    static constexpr int PATCH_PIECES = (WNPIX * CPP + 63) / 64;        // 39 / 20
    static constexpr int PATCH_BYTES = PATCH_PIECES * 1024;
    static constexpr int W_PIECES = 9 * KS * 2 * NW;                    // 36
    static constexpr int STAGE_BYTES = PATCH_BYTES + W_PIECES * 1024;   // 76 800 / 57 344
    static constexpr int PIECES = PATCH_PIECES + W_PIECES;              // 75 / 56
    static constexpr int DW = 4;                                        // waves that issue the DMA pieces (0 .. 3: one per SIMD, the older of its pair)
    static constexpr int SLOTS = (PIECES + DW - 1) / DW;                // DMA pieces per issuing wave and stage: 19 / 14
    static constexpr int PSLOTS = (PATCH_PIECES + DW - 1) / DW;         // ... of which patch pieces (per-lane source offsets): 10 / 5
    static constexpr int PER_TAP = (SLOTS + 6) / 7;                     // issued over the first seven taps: 3 / 2 per tap
    // the launch's very first stage starts computing when the patch and the weights of taps 0 .. 2 have landed (pieces are issued patch
    // first, weights in tap order): the first PRO_SLOTS slots of every issuing wave; the rest is waited for in front of tap 3
    static constexpr int PRO_SLOTS = (PATCH_PIECES + 3 * (W_PIECES / 9) + DW - 1) / DW;        // 13 / 8
    static constexpr int FMASK = CPP - 1;
    static constexpr int FSHIFT = KS == 2 ? 2 : 3;                      // swizzle key of pixel P: (P >> FSHIFT) & FMASK
    static constexpr size_t LDS = 2 * (size_t)STAGE_BYTES + 256;        // two stage buffers + the layer's 64 biases: 153 856 / 114 944
};

// every 16-lane group of the activation ds_read_b128 of every (tile row, tap) must touch 16 distinct 16-B bank quads
template <int KS>
constexpr bool wide_swizzle_ok() {
    using G = WideGeom<KS, 2>;
    constexpr int WTH = G::WTH;
    constexpr int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                   {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    for (int y = 0; y < WTH; ++y)
        for (int tap = 0; tap < 9; ++tap)
            for (int ks = 0; ks < KS; ++ks)
                for (int g = 0; g < 4; ++g) {
                    unsigned seen = 0;
                    for (int i = 0; i < 16; ++i) {
                        const int lane = groups[g][i], li = lane & 31, lh = lane >> 5;
                        const int P = (y + tap / 3) * WPW + li + tap % 3;
                        const int j = (KS == 2 ? 2 * ks + lh : lh) ^ ((P >> G::FSHIFT) & G::FMASK);
                        const int quad = ((P * G::PIX_BYTES + 16 * j) >> 4) & 15;
                        if (seen & (1u << quad)) return false;
                        seen |= 1u << quad;
                    }
                }
    return true;
}
static_assert(wide_swizzle_ok<2>() && wide_swizzle_ok<1>(), "activation reads of the wide dense kernel must be bank-conflict-free");

struct DenseWideP {
    DenseH16P d;
    int tiles_x, tiles_per_img, n_items;        // 16x32 tiles per image row / per image; (image, tile) items of the launch
    unsigned wf_bytes;                           // bytes of one weight fragment array (2 * nks KB)
};

#ifdef CIAOSR_PROBE      // developer probe build (make probe; tools/dense_wide_probe.py): cycle stamps of the last launch's workgroups, wave 0
__device__ unsigned long long g_dwprobe[256 * 8];
#define DWP_NOW() __builtin_readcyclecounter()
#define DWP_ADD(slot, t0) do { if (threadIdx.x == 0) dwacc[slot] += __builtin_readcyclecounter() - (t0); } while (0)
#else
#define DWP_NOW() 0ull
#define DWP_ADD(slot, t0) do { (void)(t0); } while (0)
#endif

template <bool LO, int R>
__global__ __launch_bounds__(512) void dense_h16_wide_kernel(DenseWideP pp) {
    constexpr int KS = LO ? 1 : 2;
    using G = WideGeom<KS, R>;
    constexpr int WTH = G::WTH, WNPIX = G::WNPIX;
    constexpr int NW = G::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const DenseH16P& p = pp.d;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, lh = lane >> 5;
    const unsigned img_bytes = (unsigned)((size_t)p.H * p.W * p.ldxb * 2);
    const i32x4 desc_x = {(int)(unsigned)(size_t)p.xb, (int)(((size_t)p.xb >> 32) & 0xFFFFu), (int)p.xb_bytes, 0x00020000};
    const i32x4 desc_w = {(int)(unsigned)(size_t)p.wf, (int)(((size_t)p.wf >> 32) & 0xFFFFu), (int)pp.wf_bytes, 0x00020000};
    const void* wlo = LO ? (const void*)p.wf_lo : (const void*)p.wf;
    const i32x4 desc_wl = {(int)(unsigned)(size_t)wlo, (int)(((size_t)wlo >> 32) & 0xFFFFu), (int)pp.wf_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int NST = p.groups * (4 / KS);         // stages of the layer
    const int kpt = 4 * p.groups;                // k16 steps per tap

    auto dma = [&](unsigned dst, unsigned voff, const i32x4& desc) {
        unsigned keep;
        const unsigned sdst = __builtin_amdgcn_readfirstlane(dst);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sdst), "s"(desc) : "memory");
    };
    // per-lane source offsets of this wave's patch pieces for the tile at (ty0, tx0): piece q = w + 4 s (waves 0 .. 3) fills LDS chunks 64 q .. 64 q + 63
    // = (pixel c / CPP, slot c % CPP), and slot holds the pixel's channel chunk slot ^ key(pixel)
    unsigned goff[G::PSLOTS];
    auto set_goff = [&](int ty0, int tx0) {
#pragma unroll
        for (int s = 0; s < G::PSLOTS; ++s) {
            const int c = 64 * (w + G::DW * s) + lane;
            const int P = c / G::CPP, slot = c % G::CPP;
            const int py = P / WPW, px = P - py * WPW;
            const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
            const bool ok = P < WNPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const int j = slot ^ ((P >> G::FSHIFT) & G::FMASK);
            goff[s] = ok ? ((unsigned)(gy * p.W + gx) * (unsigned)p.ldxb * 2u + (unsigned)j * 16u) : kOobD;
        }
    };
    // DMA piece `s` of this wave for physical stage hg of image img into stage buffer buf
    auto dma_slot = [&](int s, int buf, int hg, int img) {
        const int q = w + G::DW * s;                          // wave-uniform
        const unsigned base = lds0 + (unsigned)buf * (unsigned)G::STAGE_BYTES;
        if (s < G::PSLOTS && q < G::PATCH_PIECES) {
            const unsigned add = (unsigned)hg * (unsigned)G::PIX_BYTES + (unsigned)img * img_bytes;
            dma(base + 1024u * (unsigned)q, goff[s < G::PSLOTS ? s : 0] == kOobD ? kOobD : goff[s < G::PSLOTS ? s : 0] + add, desc_x);
        } else if (q >= G::PATCH_PIECES && q < G::PIECES) {
            const int wp = q - G::PATCH_PIECES;               // ((tap KS + ks) 2 + nt) NW + lo
            const int lo = wp % NW, t2 = wp / NW, nt = t2 & 1, t3 = t2 >> 1, ks = t3 % KS, tap = t3 / KS;
            const unsigned fi = (unsigned)(nt * p.nks + tap * kpt + KS * hg + ks);
            dma(base + (unsigned)G::PATCH_BYTES + 1024u * (unsigned)wp, fi * 1024u + (unsigned)lane * 16u, lo ? desc_wl : desc_w);
        }
    };

    // activation fragment addresses (tile-relative, the same for every item): tile row yr = 2 w + r, tap -> byte offset in a stage buffer
    int baddr[R][9];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int P = (R * w + r + tap / 3) * WPW + li + tap % 3;
            baddr[r][tap] = P * G::PIX_BYTES + ((lh ^ ((P >> G::FSHIFT) & G::FMASK)) << 4);
        }
    const int waddr = G::PATCH_BYTES + lane * 16;

    f32x16 acc[2][R];                             // [nt][r]
    float* lbias = reinterpret_cast<float*>(lds + 2 * G::STAGE_BYTES);      // the layer's 64 biases behind the stage buffers (an epilogue that
    if (t < 64) lbias[t] = p.bias[t];                                       // fetched them from L2 started with a dependent round trip per tile row)
#ifdef CIAOSR_PROBE
    unsigned long long dwacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // 0 compute loop, 1 vmcnt wait, 2 barrier, 3 epilogue, 4 lifetime, 5 stages, 6 items, 7 prologue
#endif
    const unsigned long long dw_t0 = DWP_NOW();
    const int grid = (int)gridDim.x;
    int item = (int)blockIdx.x;
    if (item >= pp.n_items) return;
    int S = 0;                                    // stages done: buffer parity
    bool epi_sync = false;                        // an epilogue's LDS transposes are not yet fenced from the next DMA pieces (workgroup-uniform)
    {   // very first stage of this workgroup
        const int tl = item % pp.tiles_per_img;
        set_goff((tl / pp.tiles_x) * WTH, (tl % pp.tiles_x) * WTW);
        const int rot0 = (((tl / pp.tiles_x) * WTH / 16) * pp.tiles_x + tl % pp.tiles_x) % NST;     // index of the 16 x 32 parent tile
        if (w < G::DW) {
#pragma unroll
            for (int s = 0; s < G::SLOTS; ++s) dma_slot(s, 0, rot0, item / pp.tiles_per_img);
        }
        // every CU pulls its first 75 KB at the same time (19 MB chip-wide: ~5 us).  Taps 0 .. 2 need the patch and a third of the weights:
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(G::SLOTS - G::PRO_SLOTS) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        DWP_ADD(7, dw_t0);
    }
    bool pro_tail = true;                         // the rest of the first stage's weights is still in flight (workgroup-uniform)
#pragma unroll 1
    for (; item < pp.n_items; item += grid) {
        const int img = item / pp.tiles_per_img, tl = item - img * pp.tiles_per_img;
        const int ty0 = (tl / pp.tiles_x) * WTH, tx0 = (tl % pp.tiles_x) * WTW;
        const int rot = ((ty0 / 16) * pp.tiles_x + tl % pp.tiles_x) % NST;       // by the 16 x 32 parent tile: the same in both tile shapes
        const int nitem = item + grid;
        const bool has_next = nitem < pp.n_items;
        const int nimg = has_next ? nitem / pp.tiles_per_img : 0, ntl = has_next ? nitem - nimg * pp.tiles_per_img : 0;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][r][e] = 0.f;
#pragma unroll 1
        for (int st = 0; st < NST; ++st) {
            const bool last = st + 1 == NST;
            if (last && has_next) set_goff((ntl / pp.tiles_x) * WTH, (ntl % pp.tiles_x) * WTW);     // the DMAs of this stage fetch the next item
            const bool more = !last || has_next;
            int nhg = st + 1 + rot;                                   // physical stage the DMAs of this stage fetch
            if (nhg >= NST) nhg -= NST;
            if (last) nhg = has_next ? (((ntl / pp.tiles_x) * WTH / 16) * pp.tiles_x + ntl % pp.tiles_x) % NST : 0;
            const int nim = last ? nimg : img;
            const int buf = S & 1;
            const unsigned char* pb = lds + buf * G::STAGE_BYTES;
            uint4 a[3][2][NW], b[3][R];                               // [set][nt][weight set], [set][r]: operands are read TWO steps ahead
            auto load_ab = [&](int set, int tap, int ks) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int lo = 0; lo < NW; ++lo)
                        a[set][nt][lo] = *reinterpret_cast<const uint4*>(pb + waddr + (((tap * KS + ks) * 2 + nt) * NW + lo) * 1024);
#pragma unroll
                for (int r = 0; r < R; ++r) b[set][r] = *reinterpret_cast<const uint4*>(pb + (baddr[r][tap] ^ (ks << 5)));
            };
            // (with eight waves on the LDS port a ds_read_b128 takes longer than the 4 MFMAs of one step: one step of lookahead left the
            // waves parked in s_waitcnt for 41 % of their cycles, SQ_WAIT_ANY)
            const unsigned long long dw_s0 = DWP_NOW();
            load_ab(0, 0, 0);
            load_ab(1, 1 / KS, 1 % KS);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int idx = tap * KS + ks, cur = idx % 3;
                    if (idx + 2 == 3 * KS && pro_tail) {
                        // first stage of the launch: the operands of tap 3 are READ here (two steps ahead).  Behind the prologue's last pieces
                        // this wave has issued the next stage's pieces of the taps it has finished: those may stay in flight
                        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(((3 * KS - 2) / KS) * G::PER_TAP) : "memory");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                        pro_tail = false;
                    }
                    if (idx + 2 < 9 * KS) load_ab((idx + 2) % 3, (idx + 2) / KS, (idx + 2) % KS);
#pragma unroll
                    for (int lo = 0; lo < NW; ++lo)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int r = 0; r < R; ++r) acc[nt][r] = mfma_h16<kF16>(a[cur][nt][lo], b[cur][r], acc[nt][r]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // The DMA pieces of the next stage are issued by waves 0 .. 3 ONLY -- one wave per SIMD, the older of its pair -- two or three
                // per tap over the first seven taps (they then have the rest of the stage to land).  A piece costs its wave 60-180 cycles of
                // issue during which it feeds no MFMA; the probe of the first cut (every wave issuing its share, tools/dense_wide_probe.py)
                // showed both waves of a SIMD stuck in their DMA issue at the same time: the older took 47 cycles per MFMA, the younger was
                // starved and then ran ALONE at 30.5 -- 5590 cycles per stage against the 4608 of MFMA issue.  Now the younger wave of every
                // SIMD is pure MFMA + LDS reads and fills the pipe whenever the older one is busy issuing.
                // the barrier between the previous item's epilogue transposes and the first DMA piece into their buffer sits HERE, behind the
                // first tap's MFMAs of the next item: a wave that leaves its epilogue early feeds the matrix pipe instead of waiting
                if (tap == 0 && epi_sync) { __syncthreads(); epi_sync = false; }
                if (more && w < G::DW) {
#pragma unroll
                    for (int e = 0; e < G::PER_TAP; ++e)
                        if (G::PER_TAP * tap + e < G::SLOTS) dma_slot(G::PER_TAP * tap + e < G::SLOTS ? G::PER_TAP * tap + e : 0, buf ^ 1, nhg, nim);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            DWP_ADD(0, dw_s0);
            const unsigned long long dw_s1 = DWP_NOW();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DWP_ADD(1, dw_s1);
            const unsigned long long dw_s2 = DWP_NOW();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            DWP_ADD(2, dw_s2);
#ifdef CIAOSR_PROBE
            dwacc[5] += 1;
#endif
            ++S;
        }
        const unsigned long long dw_e0 = DWP_NOW();
        // ---- epilogue of this item: the stage buffer just computed from is free (the other one holds the next item's first stage);
        // this wave's 4-KB region of it turns the accumulator layout (lane = pixel, 4 consecutive channels per quad) into pixel rows
        {
            unsigned char* tb = lds + ((S - 1) & 1) * G::STAGE_BYTES + w * 4096;
            const int pr0 = lane >> 3, c8 = lane & 7;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = ty0 + R * w + r;
                const bool y_ok = y < p.H;
                // 16-bit copy: [32 pixels][64 channels x 2 B], 16-B chunks swizzled with the pixel
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bb = *reinterpret_cast<const float4*>(lbias + 32 * nt + 8 * q + 4 * lh);
                        const float v0 = fmaxf(acc[nt][r][4 * q] + bb.x, 0.f), v1 = fmaxf(acc[nt][r][4 * q + 1] + bb.y, 0.f);
                        const float v2 = fmaxf(acc[nt][r][4 * q + 2] + bb.z, 0.f), v3 = fmaxf(acc[nt][r][4 * q + 3] + bb.w, 0.f);
                        acc[nt][r][4 * q] = v0; acc[nt][r][4 * q + 1] = v1; acc[nt][r][4 * q + 2] = v2; acc[nt][r][4 * q + 3] = v3;
                        *reinterpret_cast<uint2*>(tb + li * 128 + (((4 * nt + q) ^ (li & 7)) << 4) + 8 * lh) = pack_h16x4<kF16>(v0, v1, v2, v3);
                    }
                wave_lds_sync();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pr = 8 * j + pr0, x = tx0 + pr;
                    const uint4 v = *reinterpret_cast<const uint4*>(tb + pr * 128 + ((c8 ^ (pr & 7)) << 4));
                    if (y_ok && x < p.W)
                        *reinterpret_cast<uint4*>(p.xb_out + ((size_t)img * p.H * p.W + (size_t)y * p.W + x) * p.ldxb + p.col_out + 8 * c8) = v;
                }
                wave_lds_sync();
                if (p.x) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4*>(tb + li * 128 + (((2 * q + lh) ^ (li & 7)) << 4)) =
                                make_float4(acc[nt][r][4 * q], acc[nt][r][4 * q + 1], acc[nt][r][4 * q + 2], acc[nt][r][4 * q + 3]);
                        wave_lds_sync();
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int pr = 8 * j + pr0, x = tx0 + pr;
                            const float4 v = *reinterpret_cast<const float4*>(tb + pr * 128 + ((c8 ^ (pr & 7)) << 4));
                            if (y_ok && x < p.W)
                                *reinterpret_cast<float4*>(p.x + ((size_t)img * p.H * p.W + (size_t)y * p.W + x) * p.ldx + p.col_out + 32 * nt + 4 * c8) = v;
                        }
                        wave_lds_sync();
                    }
                }
            }
        }
        epi_sync = has_next;                      // the next stage's DMA pieces land in the buffer the transposes used: barrier in front of them (below)
        DWP_ADD(3, dw_e0);
#ifdef CIAOSR_PROBE
        dwacc[6] += 1;
#endif
    }
#ifdef CIAOSR_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 256) {
        dwacc[4] = __builtin_readcyclecounter() - dw_t0;
        for (int i = 0; i < 8; ++i) g_dwprobe[blockIdx.x * 8 + i] = dwacc[i];
    }
#endif
}

bool dense_h16_wide_ok(int H, int W) { return ceil_div(H, 16) * ceil_div(W, WTW) >= 32; }

// fp32 columns [col, col+64) of X -> the same columns of the bf16 copy
__global__ void cast_group_h16_kernel(const float* __restrict__ X, int ldx, unsigned short* __restrict__ Xb, int ldxb, int col,
                                       long HW) {
    const long n = HW * 16;                   // float4 units
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long pix = i >> 4;
        const int c = (int)(i & 15) * 4;
        const float4 v = *reinterpret_cast<const float4*>(X + pix * ldx + col + c);
        *reinterpret_cast<uint2*>(Xb + pix * ldxb + col + c) =
            pack_h16x4<kF16>(v.x, v.y, v.z, v.w);
    }
}

int dense_h16_tiles(int H, int W) { return ceil_div(H, DT) * ceil_div(W, DT); }

int cast_group_h16(const float* X, int ldx, unsigned short* Xb, int ldxb, int col, long HW, hipStream_t s) {
    ProfScope prof("enc_cast" CIAOSR_H16_SUFFIX, s);
    const long n = HW * 16;
    int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL(cast_group_h16_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, s, X, ldx, Xb, ldxb, col, HW);
    return launch_status("cast_group" CIAOSR_H16_SUFFIX);
}

// dense layer l of a block: input groups 0..l of Xb, output group l+1 (fp32 into X, bf16 into Xb); n_img images back to back
// route: 0 = automatic (the 16x32-tile persistent kernel of round 6 from 32 such tiles per image on, else the 12x12-tile kernels),
// 1 = the 12x12-tile kernels whatever the map size (developer A/B and the parity test of the two cuts).  Chosen by the size of ONE image:
// a tile's result does not depend on the batch it is computed in.
int dense_layer_h16(float* X, int ldx, unsigned short* Xb, int ldxb, int H, int W, int l, const void* frag16, const void* frag16_lo,
                     const float* bias, int n_img, hipStream_t s, int route) {
    CIAOSR_CHECK_ARG(Xb && frag16 && bias && (ldx & 3) == 0 && (ldxb & 7) == 0);      // X null: no fp32 copy of the output
    const size_t xb_bytes = (size_t)n_img * H * W * ldxb * 2;
    CIAOSR_CHECK_ARG(n_img >= 1 && xb_bytes < 0xFFFFFF00ull);
    DenseH16P p;
    p.xb = Xb; p.ldxb = ldxb; p.xb_bytes = (unsigned)xb_bytes;
    p.H = H; p.W = W; p.tiles_x = ceil_div(W, DT);
    p.n_img = n_img;
    p.groups = l + 1;
    p.wf = reinterpret_cast<const uint4*>(frag16); p.nks = 9 * 64 * (l + 1) / 16;
    p.wf_lo = reinterpret_cast<const uint4*>(frag16_lo);
    p.bias = bias;
    p.x = X; p.ldx = ldx; p.xb_out = Xb; p.col_out = 64 * (l + 1);
    CIAOSR_BIG_LDS(dense_h16_kernel<true>, kDenseLds);
    CIAOSR_BIG_LDS(dense_h16_kernel<false>, kDenseLds);
    CIAOSR_BIG_LDS(dense_h16_dma_kernel, kDense2Lds);
    ProfScope prof("enc_dense" CIAOSR_H16_SUFFIX, s);
    if (route != 1 && dense_h16_wide_ok(H, W)) {
        DenseWideP wp;
        wp.d = p;
        wp.tiles_x = ceil_div(W, WTW);
        // tile shape by the size of the LAUNCH (bitwise the same outputs, see WideGeom): rounds of the 256 CUs x relative cost of a round
        const long n16 = (long)wp.tiles_x * ceil_div(H, 16) * n_img, n8 = (long)wp.tiles_x * ceil_div(H, 8) * n_img;
        const bool small = 1.2 * (double)((n8 + 255) / 256) < 2.0 * (double)((n16 + 255) / 256);
        wp.tiles_per_img = wp.tiles_x * ceil_div(H, small ? 8 : 16);
        wp.n_items = wp.tiles_per_img * n_img;
        wp.wf_bytes = (unsigned)((size_t)2 * p.nks * 1024);
        const int grid = wp.n_items < 256 ? wp.n_items : 256;        // persistent: one workgroup per CU
#define CIAOSR_LAUNCH_WIDE(LO_, R_)                                                                                              \
        do {                                                                                                                     \
            CIAOSR_BIG_LDS((dense_h16_wide_kernel<LO_, R_>), (WideGeom<(LO_) ? 1 : 2, R_>::LDS));                                \
            hipLaunchKernelGGL((dense_h16_wide_kernel<LO_, R_>), dim3(grid), dim3(512), (WideGeom<(LO_) ? 1 : 2, R_>::LDS), s, wp); \
        } while (0)
        if (p.wf_lo) { if (small) CIAOSR_LAUNCH_WIDE(true, 1); else CIAOSR_LAUNCH_WIDE(true, 2); }
        else { if (small) CIAOSR_LAUNCH_WIDE(false, 1); else CIAOSR_LAUNCH_WIDE(false, 2); }
#undef CIAOSR_LAUNCH_WIDE
        return launch_status("dense_wide" CIAOSR_H16_SUFFIX);
    }
    // no process-global switches in the product library: the developer A/B knobs exist in the CIAOSR_PROBE build only
#ifdef CIAOSR_PROBE
    static const int variant = getenv("CIAOSR_DENSE_V") ? atoi(getenv("CIAOSR_DENSE_V")) : 2;     // 1 = the round-2 kernel
    static const int slices_env = getenv("CIAOSR_DENSE_SLICES") ? atoi(getenv("CIAOSR_DENSE_SLICES")) : 2;
    const int slices = slices_env < 1 ? 1 : slices_env;
#else
    constexpr int variant = 2, slices = 2;
#endif
    if (p.wf_lo)
        hipLaunchKernelGGL(dense_h16_kernel<true>, dim3(dense_h16_tiles(H, W)), dim3(256), kDenseLds, s, p);
    else if (variant == 1 || n_img < 2)      // one image = 256 workgroups on 256 CUs: nothing to pair up, the deeper pipeline of the first kernel wins (2.74 vs 2.96 ms)
        hipLaunchKernelGGL(dense_h16_kernel<false>, dim3(dense_h16_tiles(H, W)), dim3(256), kDenseLds, s, p);
    else {   // two workgroups per CU; a batch is cut into image slices so that both slots of every CU are filled
        hipLaunchKernelGGL(dense_h16_dma_kernel, dim3(dense_h16_tiles(H, W), n_img < slices ? n_img : slices), dim3(256), kDense2Lds, s, p);
    }
    return launch_status("dense" CIAOSR_H16_SUFFIX);
}

}  // namespace CIAOSR_H16_NS
}  // namespace ciaosr

#ifdef CIAOSR_PROBE
#if CIAOSR_F16
extern "C" int ciaosr_debug_probe_dw_read(unsigned long long* host, int n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ciaosr::f16::g_dwprobe), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
#endif
#endif
