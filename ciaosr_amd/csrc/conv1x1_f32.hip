// Weights-resident exact-fp32 1x1 convolution for tall, narrow problems: out[m][0..64) = sum_k X[m][k] W[n][k] + bias + res
// with N = 64 output channels and K <= 576 input channels -- the local feature fusion of a residual dense block
// (mmedit RDB.lff, called from ciaosr_net.py:337: 576 -> 64 channels on every LR pixel, 16 times per tile).
//
// The generic implicit-GEMM tile (conv_gemm_kernel<64,64>, conv_f32.hip) stages both operands through LDS per 64-row tile and
// ran this shape at 0.38 of the fp32 MFMA peak while moving only 2.7 TB/s: with N = 64 there is a single column tile, so every
// k-stage pays a barrier for 64 x 64 x 32 MACs.  Here the WEIGHTS are the resident operand -- [64][K] fp32 in LDS (148 KB, one
// 512-thread workgroup per CU, loaded once) -- and every wave streams whole 32-row tiles of X straight from memory into MFMA
// operand registers, 18 float4 (a 144-channel chunk) at a time with the next chunk in flight under the current chunk's 144
// MFMAs; no barrier after the weight load, no LDS traffic for X.  v_mfma_f32_32x32x2_f32 with swapped operands (weights = A,
// activations = B): a lane owns one row and 4 consecutive channels per accumulator quad, so the epilogue (bias, fp32 residual, two
// destinations) is float4 loads / stores.  One float4 of X feeds four MFMAs per output half because the k index inside a chunk
// may be permuted consistently on both operands: lane (i = l & 31, h = l >> 5) holds k = 8 j + 4 h + e for MFMA e of step j.
// HBM need at the full MFMA rate: 8 B/clk per CU = 4.9 TB/s: the kernel sits where the two rooflines meet.
#include "ops.h"

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int R1_CH = 18;                 // float4 steps (8 k each) per chunk
constexpr int R1_MAXK = 576;
constexpr unsigned kOobR1 = 0xFFFFFFF0u;

struct Res1x1P {
    const float* X; int ldx; unsigned x_bytes;
    const float* W; int ldw;
    const float* bias;
    const float* res; int ldres;
    float* dst; int ld_dst;
    float* dst2; int ld_dst2;
    long M;
    int K, nch, wpitch;                   // nch = chunks of 144 k; wpitch = LDS row pitch in bytes
    int n_tiles;
};

__global__ __launch_bounds__(512, 2) void conv1x1_resident_f32_kernel(Res1x1P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];        // [64][wpitch]: W rows, zero-padded to nch * 144 columns
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
    {
        const int cpr = p.nch * R1_CH * 2;                                    // float4 per LDS row
        for (int c = t; c < 64 * cpr; c += 512) {
            const int r = c / cpr, k4 = (c - r * cpr) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k4 < p.K) v = *reinterpret_cast<const float4*>(p.W + (size_t)r * p.ldw + k4);      // K % 4 == 0
            *reinterpret_cast<float4*>(wl + r * p.wpitch + k4 * 4) = v;
        }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const unsigned char* wrow = wl + li * p.wpitch + lh * 16;
    const int n_waves = (int)gridDim.x * 8;
    auto row_base = [&](int tile) -> unsigned {
        const long row = (long)tile * 32 + li;
        return (tile < p.n_tiles && row < p.M) ? (unsigned)((size_t)row * p.ldx * 4) + (unsigned)lh * 16u : kOobR1;
    };
    auto load_chunk = [&](i32x4 (&xf)[R1_CH], unsigned xbase, int c) {
#pragma unroll
        for (int j = 0; j < R1_CH; ++j) {
            const int k = (c * R1_CH + j) * 8 + 4 * lh;
            xf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (xbase != kOobR1 && k < p.K) ? (int)(xbase + (unsigned)(c * R1_CH + j) * 32u) : (int)kOobR1, 0, 0);
        }
    };
    f32x16 acc[2];
    auto mma_chunk = [&](const i32x4 (&xf)[R1_CH], int c) {
#pragma unroll
        for (int j = 0; j < R1_CH; ++j) {
            const float4 w0 = *reinterpret_cast<const float4*>(wrow + (c * R1_CH + j) * 32);
            const float4 w1 = *reinterpret_cast<const float4*>(wrow + 32 * p.wpitch + (c * R1_CH + j) * 32);
            const float x0 = __int_as_float(xf[j].x), x1 = __int_as_float(xf[j].y), x2 = __int_as_float(xf[j].z), x3 = __int_as_float(xf[j].w);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, x0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.x, x0, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, x1, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.y, x1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, x2, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.z, x2, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, x3, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.w, x3, acc[1], 0, 0, 0);
        }
    };
    auto epilogue = [&](int tile) {
        const long row = (long)tile * 32 + li;
        if (row >= p.M) return;
        // every load first (bias, residual: 16 float4 in flight), then the arithmetic and the stores: written load -> store per column
        // group, the possible aliasing of dst with res / bias makes hipcc keep that order and the eight round trips run one after the other
        float4 bq[2][4], rq[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * nt + 8 * g + 4 * lh;
                bq[nt][g] = *reinterpret_cast<const float4*>(p.bias + n);
                rq[nt][g] = p.res ? *reinterpret_cast<const float4*>(p.res + (size_t)row * p.ldres + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * nt + 8 * g + 4 * lh;
                const float4 b = bq[nt][g];
                float4 v = make_float4(acc[nt][4 * g] + b.x, acc[nt][4 * g + 1] + b.y, acc[nt][4 * g + 2] + b.z, acc[nt][4 * g + 3] + b.w);
                if (p.res) {
                    const float4 r = rq[nt][g];
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                *reinterpret_cast<float4*>(p.dst + (size_t)row * p.ld_dst + n) = v;
                if (p.dst2) *reinterpret_cast<float4*>(p.dst2 + (size_t)row * p.ld_dst2 + n) = v;
            }
    };
    // Two register sets (compile-time indices: the chunk loop runs over PAIRS; the host pads the chunk count to an even number, the
    // pad chunk's loads are out of range = zeros against zero weights).  One chunk is always in flight under the 144 MFMAs of the
    // previous one -- across tiles too: the first chunk of the wave's NEXT tile is requested before the last chunk of this one runs,
    // so only a wave's very first chunk sees the memory latency.
    i32x4 xa[R1_CH], xb[R1_CH];
    int tile = (int)blockIdx.x * 8 + w;
    unsigned xbase = row_base(tile);
    load_chunk(xa, xbase, 0);
#pragma unroll 1
    for (; tile < p.n_tiles; tile += n_waves) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
        const unsigned xnext = row_base(tile + n_waves);
#pragma unroll 1
        for (int c = 0; c < p.nch; c += 2) {
            load_chunk(xb, xbase, c + 1);
            mma_chunk(xa, c);
            if (c + 2 < p.nch) load_chunk(xa, xbase, c + 2);
            else load_chunk(xa, xnext, 0);
            mma_chunk(xb, c + 1);
        }
        epilogue(tile);
        xbase = xnext;
    }
}

bool conv1x1_resident_ok(long M, int N, int K, int ldx, int ldw) {
    // M = rows of ONE image: the choice must not depend on how many tiles share a launch (a tile's result is bitwise the same alone,
    // in a batch and on a side stream).  Big maps only; a single 192x192 tile (1152 row tiles for 2048 waves) runs it at 1.07 ms
    // against 0.82 ms of the generic tile kernel, a batch of 8 at 0.57 against 0.82 per tile.
    return N == 64 && K >= 64 && K <= R1_MAXK && (K & 7) == 0 && (ldx & 3) == 0 && (ldw & 3) == 0 && M >= 32768 &&
           ((size_t)(M - 1) * ldx + K) * 4 < 0xFFFFFF00ull;
}

// dst[m][0..64) (and dst2 when given) = X[m][0..K) . W^T + bias + res[m][0..64); rows [0, M) of every operand
int conv1x1_resident_f32(const float* X, int ldx, const float* W, int ldw, const float* bias, const float* res, int ldres, float* dst,
                         int ld_dst, float* dst2, int ld_dst2, long M, int K, hipStream_t s, const char* tag) {
    CIAOSR_CHECK_ARG(X && W && bias && dst && M > 0 && conv1x1_resident_ok(32768, 64, K, ldx, ldw) && ((size_t)(M - 1) * ldx + K) * 4 < 0xFFFFFF00ull);
    CIAOSR_CHECK_ARG((ldres & 3) == 0 && (ld_dst & 3) == 0 && (ld_dst2 & 3) == 0 && aligned16(X) && aligned16(W) && aligned16(bias) &&
                     aligned16(dst) && (!res || aligned16(res)) && (!dst2 || aligned16(dst2)));
    Res1x1P p;
    p.X = X; p.ldx = ldx; p.x_bytes = (unsigned)(((size_t)(M - 1) * ldx + K) * 4);
    p.W = W; p.ldw = ldw; p.bias = bias; p.res = res; p.ldres = ldres;
    p.dst = dst; p.ld_dst = ld_dst; p.dst2 = dst2; p.ld_dst2 = ld_dst2;
    p.M = M; p.K = K; p.nch = (ceil_div(K, R1_CH * 8) + 1) & ~1;      // even: the kernel walks chunk pairs
    p.wpitch = p.nch * R1_CH * 32 + 16;          // +16 B: the 32 rows of a ds_read_b128 lane group fall on distinct bank quads
    p.n_tiles = (int)((M + 31) / 32);
    const size_t lds = (size_t)64 * p.wpitch;
    CIAOSR_BIG_LDS(conv1x1_resident_f32_kernel, lds);
    int grid = ceil_div(p.n_tiles, 8);
    if (grid > 256) grid = 256;
    ProfScope prof(tag, s);
    hipLaunchKernelGGL(conv1x1_resident_f32_kernel, dim3(grid), dim3(512), lds, s, p);
    return launch_status("conv1x1_resident_f32");
}

}  // namespace ciaosr
