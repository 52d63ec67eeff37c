// fp32 dense-block convolution for big maps (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32): the halo-resident,
// K-sliced design of dense_h16.hip in the contract precision.
//
// One 3x3 dense layer l of a residual dense block (mmedit RDB.layers[l].conv over cat(x, d_0 .. d_{l-1}), called from
// ciaosr_net.py:330-337) in GATHER form: K = 9 * 64 (l+1), N = 64.  The tap-major implicit GEMM of conv_f32.hip
// re-fetches a 64-channel K-stage of activations AND weights for every 64x64x64 MACs and tops out at 0.60 of the MFMA
// peak on a 192x192 tile (operand stream, not MFMA).  Here:
//   * workgroup = 12x12 output pixels x all 64 output channels (a 192x192 tile = exactly 256 workgroups, one per CU);
//   * the 14x14-pixel halo patch of one 64-channel input group (fp32, 50 KB) is loaded into LDS once and serves all
//     9 taps; the next group's patch is prefetched into registers meanwhile (two LDS buffers);
//   * the 4 waves split K (wave w owns channels 16w..16w+15 of every group) and each accumulates the whole
//     160(144 used) x 64 tile = 5 x 2 MFMA tiles (160 accumulator registers): one 16-B LDS read feeds 8 MFMAs and
//     one pre-packed 1-KB weight fragment from L2 (ciaosr_pack_fragments_f32 order, no LDS) feeds 20;
//   * per group a workgroup ingests 50 KB of patch + 147 KB of weights against 46k cycles of MFMA work (4.3 B/clk/CU);
//   * the K-slices are summed through LDS once per layer, in a fixed order (deterministic).
// Same MACs as the reference convolution, fp32 products and sums; only the summation order differs from conv_f32.hip.

#include "ops.h"

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int FT = 12;                       // output tile edge (pixels)
constexpr int FP = FT + 2;                   // patch edge with the 1-pixel halo
constexpr int FPS = 272;                     // bytes per patch pixel: 64 fp32 + 16 B pad (conflict-free ds_read_b128)
constexpr int FPATCH = FP * FP * FPS;        // 53 312 B per buffer
constexpr int FCHUNKS = FP * FP * 16;        // 16-byte chunks of a patch
constexpr int FLOADS = (FCHUNKS + 255) / 256;   // 13
constexpr int FMT = 5;                       // 32-pixel MFMA tiles per workgroup (160 rows, 144 used)
constexpr size_t kDenseF32Lds = 2 * (size_t)FPATCH;   // 106 624 B >= K-slice reduction scratch (81 920 B)
constexpr unsigned kOobDF = 0xFFFFFFF0u;

struct DenseF32P {
    float* x; int ldx;                       // fp32 feature buffer [HW][ldx], 64-channel groups; read and written
    unsigned x_bytes;
    int H, W, tiles_x;
    int n_img;                               // images in the buffer (back to back); one workgroup walks its tile of every image
    int groups;                              // input groups of this layer (l + 1)
    const float4* wf; int nj;                // fragments [2][nj][64 lanes] float4, nj = 9*cin/8
    const float* bias;                       // [64]
    int col_out;
};

__global__ __launch_bounds__(256) void dense_f32_kernel(DenseF32P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsf[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
    const int ty0 = (blockIdx.x / p.tiles_x) * FT, tx0 = (blockIdx.x % p.tiles_x) * FT;
    // The workgroup walks its 12x12 tile of EVERY image of the batch (same weights, own rows) as one software pipeline: the first
    // weights and the halo patch of image i + 1 are requested during the last input group of image i, so only the first image pays
    // the cold start.  Per image the work, its order and hence the result are those of a single-image launch.
    const unsigned img_bytes = (unsigned)((size_t)p.H * p.W * p.ldx * 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.x, 0, p.x_bytes, 0x00020000);

    // patch staging: thread -> 16-byte chunks t + 256 s  (pixel = chunk / 16, 16 chunks = 64 channels)
    unsigned goff[FLOADS];
    int loff[FLOADS];
#pragma unroll
    for (int s = 0; s < FLOADS; ++s) {
        const int c = t + 256 * s;
        const int px = c >> 4, part = c & 15;
        const int py = px / FP, pxx = px - py * FP;
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + pxx;
        const bool ok = c < FCHUNKS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        goff[s] = ok ? ((unsigned)(gy * p.W + gx) * (unsigned)p.ldx * 4u + (unsigned)part * 16u) : kOobDF;
        loff[s] = c < FCHUNKS ? px * FPS + part * 16 : -1;
    }
    i32x4 P[FLOADS];
    auto load_chunk = [&](int s, int g, int img) {
        P[s] = __builtin_amdgcn_raw_buffer_load_b128(
            rs, goff[s] == kOobDF ? (int)kOobDF : (int)(goff[s] + (unsigned)g * 256u + (unsigned)img * img_bytes), 0, 0);
    };
    auto store_patch = [&](int buf) {
#pragma unroll
        for (int s = 0; s < FLOADS; ++s)
            if (loff[s] >= 0) *reinterpret_cast<i32x4*>(ldsf + buf * FPATCH + loff[s]) = P[s];
    };

    // B operand (activations): pixel of lane li in each of the 5 pixel tiles, this wave's 16-channel K slice.
    // Offsets are biased by the most negative tap offset so that every tap offset is a non-negative immediate.
    constexpr int kTapMin = (-1 * FP - 1) * FPS;
    int poff[FMT];
#pragma unroll
    for (int r = 0; r < FMT; ++r) {
        int idx = 32 * r + li;
        idx = idx < FT * FT ? idx : FT * FT - 1;
        const int y = idx / FT, x = idx - y * FT;
        poff[r] = ((y + 1) * FP + (x + 1)) * FPS + (16 * w + 4 * lh) * 4 + kTapMin;
    }
    const float4* wl = p.wf + lane;
    const int jpt = 8 * p.groups;             // 8-deep k-chunks per tap (cin / 8)
    // fragment (nt, c) of (group g, tap): k-chunk j = tap*jpt + 8g + 2w + c
    auto frag = [&](int nt, int g, int tap, int c) -> float4 { return wl[(size_t)(nt * p.nj + tap * jpt + 8 * g + 2 * w + c) * 64]; };

    f32x16 acc[2][FMT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < FMT; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][r][e] = 0.f;
    };
    zero_acc();

#pragma unroll
    for (int s = 0; s < FLOADS; ++s) load_chunk(s, 0, 0);
    store_patch(0);
    float4 w0[2][2], w1[2][2];                // [chunk c][nt]: current tap and the next
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) w0[c][nt] = frag(nt, 0, 0, c);
    __syncthreads();

    const int G = p.groups;
    int pbuf = 0;                             // LDS patch buffer of the current (image, group)
#pragma unroll 1
    for (int img = 0; img < p.n_img; ++img) {
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const bool last_g = g + 1 == G;
        const bool more = !last_g || img + 1 < p.n_img;           // another (image, group) follows: prefetch it
        const int ng_next = last_g ? 0 : g + 1, nimg_next = last_g ? img + 1 : img;
        const unsigned char* pb = ldsf + pbuf * FPATCH;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            {   // weights of the next tap (a tap is 5120 MFMA cycles: one tap of lookahead covers any L2 latency)
                int ng = g, ntap = tap + 1;
                bool have = true;
                if (ntap == 9) { ntap = 0; ng = ng_next; have = more; }       // the next group -- of this image or the first of the next
                if (have) {
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) w1[c][nt] = frag(nt, ng, ntap, c);
                }
            }
            if (more) {                                              // next group's patch, spread over the taps
                if (tap < FLOADS) load_chunk(tap, ng_next, nimg_next);
                if (tap + 9 < FLOADS) load_chunk(tap + 9, ng_next, nimg_next);
            }
            const int toff = ((tap / 3 - 1) * FP + (tap % 3 - 1)) * FPS - kTapMin;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float4 b[FMT];
#pragma unroll
                for (int r = 0; r < FMT; ++r) b[r] = *reinterpret_cast<const float4*>(pb + poff[r] + toff + c * 32);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float4 a = w0[c][nt];
#pragma unroll
                    for (int r = 0; r < FMT; ++r) {
                        acc[nt][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[r].x, acc[nt][r], 0, 0, 0);
                        acc[nt][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[r].y, acc[nt][r], 0, 0, 0);
                        acc[nt][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[r].z, acc[nt][r], 0, 0, 0);
                        acc[nt][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[r].w, acc[nt][r], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) w0[c][nt] = w1[c][nt];
        }
        if (!last_g) {
            store_patch(pbuf ^ 1);
            pbuf ^= 1;
        }
        __syncthreads();
    }

    // K-slice reduction + epilogue of this image, one 32-channel half at a time through LDS (the scratch overlays both patch buffers:
    // the next image's patch stays in registers until it is done):
    // red[w][r][q][lane] = float4 of accumulator registers 4q..4q+3 (= channels 8q + 4lh .. +3 of pixel li of tile r)
    float* const xi = p.x + (size_t)img * p.H * p.W * p.ldx;
    float4* red = reinterpret_cast<float4*>(ldsf);
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int r = 0; r < FMT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                red[((w * FMT + r) * 4 + q) * 64 + lane] =
                    make_float4(acc[nt][r][4 * q], acc[nt][r][4 * q + 1], acc[nt][r][4 * q + 2], acc[nt][r][4 * q + 3]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < FMT; ++u) {
            const int unit = t + 256 * u;                 // (r, q, lane)
            const int ul = unit & 63, q = (unit >> 6) & 3, r = unit >> 8;
            float4 v = red[((0 * FMT + r) * 4 + q) * 64 + ul];
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) {
                const float4 o = red[((ww * FMT + r) * 4 + q) * 64 + ul];
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            const int idx = 32 * r + (ul & 31);
            const int y = ty0 + idx / FT, x = tx0 + idx % FT;
            if (idx < FT * FT && y < p.H && x < p.W) {
                const int co = 32 * nt + 8 * q + 4 * (ul >> 5);
                const float4 b = *reinterpret_cast<const float4*>(p.bias + co);
                v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f);
                v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
                *reinterpret_cast<float4*>(xi + ((size_t)y * p.W + x) * p.ldx + p.col_out + co) = v;
            }
        }
        __syncthreads();
    }
    if (img + 1 < p.n_img) {                  // next image: its first patch (requested during the last group) goes to LDS now
        zero_acc();
        pbuf = 0;
        store_patch(0);
        __syncthreads();
    }
    }
}

int dense_f32_tiles(int H, int W) { return ceil_div(H, FT) * ceil_div(W, FT); }

// dense layer l of a block: input groups 0..l of X, output group l+1; X holds n_img images of H x W rows back to back
int dense_layer_f32(float* X, int ldx, int H, int W, int l, const float* frag, const float* bias, int n_img, hipStream_t s) {
    CIAOSR_CHECK_ARG(X && frag && bias && (ldx & 3) == 0 && aligned16(X) && aligned16(frag) && aligned16(bias));
    const size_t x_bytes = (size_t)n_img * H * W * ldx * 4;
    CIAOSR_CHECK_ARG(n_img >= 1 && x_bytes < 0xFFFFFF00ull);
    DenseF32P p;
    p.x = X; p.ldx = ldx; p.x_bytes = (unsigned)x_bytes;
    p.H = H; p.W = W; p.tiles_x = ceil_div(W, FT);
    p.n_img = n_img;
    p.groups = l + 1;
    p.wf = reinterpret_cast<const float4*>(frag); p.nj = 9 * 64 * (l + 1) / 8;
    p.bias = bias;
    p.col_out = 64 * (l + 1);
    CIAOSR_BIG_LDS(dense_f32_kernel, kDenseF32Lds);
    ProfScope prof("enc_dense_gather", s);
    hipLaunchKernelGGL(dense_f32_kernel, dim3(dense_f32_tiles(H, W)), dim3(256), kDenseF32Lds, s, p);
    return launch_status("dense_f32");
}

}  // namespace ciaosr
