// Correlation scores of CrossScaleAttention (arch_csnln.py:494-500: F.conv2d of the 3x3 query patches with the L2-normalised
// 3x3 key patches, x softmax_scale) WITHOUT forming the 288-wide patch rows (round 3).
//
//   S[p][l] = alpha n_l sum_{a,b in {-1,0,1}} D[p + (a,b)][l + (a,b)],      D[p'][l'] = <M[p'], R[l']>  (C/2 = 32 channels)
// (zero outside either map; n_l = 1 / max(|3x3 patch of R at l|, floor)): the patch correlation is a 3x3 DIAGONAL BOX SUM of a
// per-pixel correlation with K = 32 instead of K = 288.  A workgroup owns 8x16 query pixels x 8x16 key pixels = 128 x 128 scores:
//   A. D over the two halos (180 x 180, padded to 192 x 192 = 6 x 6 MFMA tiles of K = 32: 9.2k MFMA cycles per wave against the
//      42k of the 128 x 128 x 288 GEMM tile) on the exact-fp32 MFMA, operands straight from L2 one tile ahead, into LDS
//      ([180][185] fp32, 133 KB: one workgroup per CU);
//   B. every thread box-sums 64 scores out of LDS (9 reads + 8 adds each; lanes along l: conflict-free rows), scales and stores.
// 4.5x fewer flops than the GEMM (halo recomputation included), the same fp32 products summed in a different order.
// Replaces csa_patch_q + csa_patch_k + the csa_scores GEMM of the fp32 path for Ch = 32.
#include "ops.h"

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BSH = 8, BSW = 16;               // block of pixels (both maps)
constexpr int BHH = BSH + 2, BHW = BSW + 2;    // with the 1-pixel halo: 10 x 18 = 180
constexpr int BHN = BHH * BHW;                 // 180
constexpr int BDP = 185;                       // D row pitch in floats (odd: column-wise accumulator writes spread over the banks)
constexpr size_t kBoxLds = (size_t)BHN * BDP * 4;      // 133 200 B
constexpr unsigned kOobB = 0xFFFFFFF0u;

struct BoxP {
    const float* M; int ldm; unsigned m_bytes; int Hp, Wp;      // query map [Hp*Wp][ldm], 32 channels
    const float* R; int ldr; unsigned r_bytes; int Hl, Wl;      // key map [Hl*Wl][ldr]
    const float* nrm;                                            // [Hl*Wl] alpha / max(patch norm, floor)
    float* S; int lds_;                                          // [Hp*Wp][lds_]
    int pbx, lbx, n_lb;                                          // blocks per row of each map, key blocks in total
};

// alpha / max(|3x3 patch|, floor) of every key pixel (zero padding)
__global__ void csa_key_norms_kernel(const float* __restrict__ R, int ldr, int Hl, int Wl, int Ch, float floor_, float alpha,
                                     float* __restrict__ nrm) {
    const int lane = threadIdx.x & 63;
    const long l = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= (long)Hl * Wl) return;
    const int ly = (int)(l / Wl), lx = (int)(l - (long)ly * Wl);
    float ss = 0.f;
    for (int e = lane; e < 9 * Ch; e += 64) {
        const int tap = e / Ch, c = e - tap * Ch;
        const int y = ly + tap / 3 - 1, x = lx + tap % 3 - 1;
        if (y >= 0 && y < Hl && x >= 0 && x < Wl) {
            const float v = R[((size_t)y * Wl + x) * ldr + c];
            ss += v * v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) nrm[l] = alpha / fmaxf(sqrtf(ss), floor_);
}

__global__ __launch_bounds__(256) void csa_scores_box_f32_kernel(BoxP p) {
    extern __shared__ __attribute__((aligned(16))) float Dl[];          // [BHN][BDP]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
    const int pb = blockIdx.x / p.n_lb, lb = blockIdx.x - pb * p.n_lb;   // key blocks fastest: a query block's M rows stay in L2
    const int py0 = (pb / p.pbx) * BSH, px0 = (pb % p.pbx) * BSW;
    const int ly0 = (lb / p.lbx) * BSH, lx0 = (lb % p.lbx) * BSW;
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.M), 0, p.m_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), 0, p.r_bytes, 0x00020000);

    // ---- A: D tiles.  Halo index h = hy * 18 + hx -> map pixel (y0 - 1 + hy, x0 - 1 + hx), zero (out-of-range offset) outside.
    // Operand fragments (32 rows x 32 channels): lane (li, lh) holds channels 8 j + 4 lh .. + 3, j = 0..3 -- the k index inside a
    // fragment may be permuted consistently on both operands.  Tile t = w + 4 k (k < 9): (query tile t / 6, key tile t % 6).
    auto row_off = [&](int h, int y0, int x0, int Hh, int Ww, int ld) -> unsigned {
        const int hy = h / BHW, hx = h - hy * BHW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        return (h < BHN && y >= 0 && y < Hh && x >= 0 && x < Ww) ? (unsigned)(((size_t)y * Ww + x) * ld * 4) + (unsigned)lh * 16u : kOobB;
    };
    auto load_tile = [&](int tl, i32x4 (&fa)[4], i32x4 (&fb)[4]) {
        const int pt = tl / 6, lt = tl - pt * 6;
        const unsigned mo = row_off(32 * pt + li, py0, px0, p.Hp, p.Wp, p.ldm);
        const unsigned ro = row_off(32 * lt + li, ly0, lx0, p.Hl, p.Wl, p.ldr);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fa[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_m, mo == kOobB ? (int)kOobB : (int)(mo + 32u * j), 0, 0);
            fb[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, ro == kOobB ? (int)kOobB : (int)(ro + 32u * j), 0, 0);
        }
    };
    i32x4 fa[2][4], fb[2][4];
    load_tile(w, fa[0], fb[0]);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int tl = w + 4 * k;
        if (k + 1 < 9) load_tile(tl + 4, fa[(k + 1) & 1], fb[(k + 1) & 1]);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // A operand = query rows (m = p'), B operand = key rows (n = l'): a lane owns key column li and 16 query rows
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i32x4 a = fa[k & 1][j], b = fb[k & 1][j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.x), __int_as_float(b.x), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.y), __int_as_float(b.y), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.z), __int_as_float(b.z), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.w), __int_as_float(b.w), acc, 0, 0, 0);
        }
        const int pt = tl / 6, lt = tl - pt * 6;
        const int lcol = 32 * lt + li;
        if (lcol < BHN) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int prow = 32 * pt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (prow < BHN) Dl[prow * BDP + lcol] = acc[r];
            }
        }
    }
    __syncthreads();

    // ---- B: box sums.  Thread -> key pixel l = t % 128 of the block, query pixels 64 (t / 128) .. + 63
    const int ll = t & 127, lyy = ll >> 4, lxx = ll & 15;
    const int ly = ly0 + lyy, lx = lx0 + lxx;
    const bool l_ok = ly < p.Hl && lx < p.Wl;
    int lcol9[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) lcol9[3 * a + b] = (lyy + a) * BHW + lxx + b;         // halo index of l + (a - 1, b - 1)
    const float sc = l_ok ? p.nrm[(size_t)ly * p.Wl + lx] : 0.f;
    const int q0 = 64 * (t >> 7);
#pragma unroll 4
    for (int q = 0; q < 64; ++q) {
        const int pl = q0 + q, pyy = pl >> 4, pxx = pl & 15;             // wave-uniform
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) s += Dl[((pyy + a) * BHW + pxx + b) * BDP + lcol9[3 * a + b]];
        const int py = py0 + pyy, px = px0 + pxx;
        if (l_ok && py < p.Hp && px < p.Wp) p.S[((size_t)py * p.Wp + px) * p.lds_ + (size_t)ly * p.Wl + lx] = s * sc;
    }
}

bool csa_scores_box_ok(int Ch, int ldm, int ldr) { return Ch == 32 && (ldm & 3) == 0 && (ldr & 3) == 0; }

// S[p][l] (row stride ld_s) = alpha <3x3 patch of M at p, 3x3 patch of R at l> / max(|patch of R at l|, floor); nrm: Hl*Wl floats scratch
int csa_scores_box_f32(const float* M, int ldm, int Hp, int Wp, const float* R, int ldr, int Hl, int Wl, int Ch, float alpha, float floor_,
                       float* nrm, float* S, int ld_s, hipStream_t s) {
    CIAOSR_CHECK_ARG(M && R && nrm && S && csa_scores_box_ok(Ch, ldm, ldr) && aligned16(M) && aligned16(R));
    const size_t mb = (size_t)Hp * Wp * ldm * 4, rb = (size_t)Hl * Wl * ldr * 4;
    CIAOSR_CHECK_ARG(mb < 0xFFFFFF00ull && rb < 0xFFFFFF00ull);
    {
        ProfScope prof("csa_key_norms", s);
        hipLaunchKernelGGL(csa_key_norms_kernel, dim3(ceil_div((long)Hl * Wl, 4)), dim3(256), 0, s, R, ldr, Hl, Wl, Ch, floor_, alpha, nrm);
    }
    int rc = launch_status("csa_key_norms");
    if (rc != CIAOSR_OK) return rc;
    BoxP p;
    p.M = M; p.ldm = ldm; p.m_bytes = (unsigned)mb; p.Hp = Hp; p.Wp = Wp;
    p.R = R; p.ldr = ldr; p.r_bytes = (unsigned)rb; p.Hl = Hl; p.Wl = Wl;
    p.nrm = nrm; p.S = S; p.lds_ = ld_s;
    p.pbx = ceil_div(Wp, BSW); p.lbx = ceil_div(Wl, BSW);
    p.n_lb = ceil_div(Hl, BSH) * p.lbx;
    const int n_pb = ceil_div(Hp, BSH) * p.pbx;
    CIAOSR_BIG_LDS(csa_scores_box_f32_kernel, kBoxLds);
    ProfScope prof("csa_scores", s);
    hipLaunchKernelGGL(csa_scores_box_f32_kernel, dim3((unsigned)n_pb * (unsigned)p.n_lb), dim3(256), kBoxLds, s, p);
    return launch_status("csa_scores_box_f32");
}

}  // namespace ciaosr
