// Correlation scores of CrossScaleAttention (arch_csnln.py:494-500: F.conv2d of the 3x3 query patches with the L2-normalised
// 3x3 key patches, x softmax_scale) WITHOUT forming the 288-wide patch rows (round 3).
//
//   S[p][l] = alpha n_l sum_{a,b in {-1,0,1}} D[p + (a,b)][l + (a,b)],      D[p'][l'] = <M[p'], R[l']>  (C/2 = 32 channels)
// (zero outside either map; n_l = 1 / max(|3x3 patch of R at l|, floor)): the patch correlation is a 3x3 DIAGONAL BOX SUM of a
// per-pixel correlation with K = 32 instead of K = 288.  An item = 8x16 query pixels x 4x16 key pixels = 128 x 64 scores:
//   A. D over the two halos (180 x 108, padded to 6 x 4 MFMA tiles of K = 32) on the exact-fp32 MFMA into LDS ([180][109] fp32);
//   B. every thread box-sums 32 scores out of LDS (9 reads + 8 adds each; lanes along l: conflict-free rows), scales and stores.
// A and B of consecutive items overlap: producer and consumer waves, two D buffers (kernel comment).  3.4x fewer flops than the GEMM
// (halo recomputation and tile padding included), the same fp32 products summed in a different order.
// Replaces csa_patch_q + csa_patch_k + the csa_scores GEMM of the fp32 path for Ch = 32.
#include "ops.h"

namespace ciaosr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BSH = 8, BSW = 16;               // block of query pixels
constexpr int BLH = 4, BLW = 16;               // block of key pixels
constexpr int BHW = BSW + 2;                   // halo row width of both blocks (18)
constexpr int BPN = (BSH + 2) * BHW;           // 180 query halo pixels (6 MFMA row tiles)
constexpr int BLN = (BLH + 2) * BHW;           // 108 key halo pixels (4 MFMA column tiles)
constexpr int BDP = 109;                       // D row pitch in floats (odd: column-wise accumulator writes spread over the banks)
constexpr int BDB = BPN * BDP * 4;             // bytes of one D buffer (78 480)
constexpr size_t kBoxLds = 2 * (size_t)BDB;    // two buffers: 156 960 B, one 512-thread workgroup per CU
constexpr unsigned kOobB = 0xFFFFFFF0u;

struct BoxP {
    const float* M; int ldm; unsigned m_bytes; int Hp, Wp;      // query map [Hp*Wp][ldm], 32 channels
    const float* R; int ldr; unsigned r_bytes; int Hl, Wl;      // key map [Hl*Wl][ldr]
    const float* nrm;                                            // [Hl*Wl] alpha / max(patch norm, floor)
    float* S; int lds_; size_t s_floats;                         // [Hp*Wp][lds_]; s_floats: extent
    int pbx, lbx, n_lb, n_items;                                 // blocks per row of each map, key blocks in total, (query, key) block pairs
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

// PERSISTENT, wave-specialised: a workgroup walks a contiguous run of (query block, key block) items.  Waves 0..3 (one per SIMD)
// PRODUCE the D block of item i + 1 on the MFMA into one LDS buffer while waves 4..7 CONSUME item i from the other: box sums, scale,
// store.  MFMA work and LDS / VALU work overlap inside the CU; one barrier per item.  A producer wave owns one 32-column key tile
// and keeps the SIX query-halo operand tiles in registers for as long as the query block does not change (144 key blocks per query
// block on the 192x192 tile), so an item costs it four 16-byte loads and 96 MFMAs.
__global__ __launch_bounds__(512) void csa_scores_box_f32_kernel(BoxP p) {
    extern __shared__ __attribute__((aligned(16))) float Dl[];          // [2][BPN][BDP]
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, lh = lane >> 5;
    const int per = (p.n_items + (int)gridDim.x - 1) / (int)gridDim.x;
    const int i0 = (int)blockIdx.x * per, i1 = min(i0 + per, p.n_items);
    if (i0 >= i1) return;
    auto item_pb = [&](int it) -> int { return it / p.n_lb; };
    auto item_lb = [&](int it) -> int { return it - (it / p.n_lb) * p.n_lb; };

    if (wv < 4) {
        // ---- producers.  Halo index h = hy * 18 + hx -> map pixel (y0 - 1 + hy, x0 - 1 + hx), zero (out-of-range offset) outside.
        // Operand fragments (32 rows x 32 channels): lane (li, lh) holds channels 8 j + 4 lh .. + 3, j = 0..3 -- the k index inside a
        // fragment may be permuted consistently on both operands.
        const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.M), 0, p.m_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), 0, p.r_bytes, 0x00020000);
        auto row_off = [&](int h, int hn, int y0, int x0, int Hh, int Ww, int ld) -> unsigned {
            const int hy = h / BHW, hx = h - hy * BHW;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            return (h < hn && y >= 0 && y < Hh && x >= 0 && x < Ww) ? (unsigned)(((size_t)y * Ww + x) * ld * 4) + (unsigned)lh * 16u : kOobB;
        };
        i32x4 fa[6][4];                                              // query-halo tiles: resident per query block
        i32x4 fbc[4], fbn[4];                                        // this wave's key-halo tile of the current / next item
        auto load_a = [&](int pb) {
            const int py0 = (pb / p.pbx) * BSH, px0 = (pb % p.pbx) * BSW;
#pragma unroll
            for (int pt = 0; pt < 6; ++pt) {
                const unsigned mo = row_off(32 * pt + li, BPN, py0, px0, p.Hp, p.Wp, p.ldm);
#pragma unroll
                for (int j = 0; j < 4; ++j) fa[pt][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_m, mo == kOobB ? (int)kOobB : (int)(mo + 32u * j), 0, 0);
            }
        };
        auto load_b = [&](int lb, i32x4 (&f)[4]) {
            const int ly0 = (lb / p.lbx) * BLH, lx0 = (lb % p.lbx) * BLW;
            const unsigned ro = row_off(32 * wv + li, BLN, ly0, lx0, p.Hl, p.Wl, p.ldr);
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, ro == kOobB ? (int)kOobB : (int)(ro + 32u * j), 0, 0);
        };
        int cur_pb = item_pb(i0);
        load_a(cur_pb);
        load_b(item_lb(i0), fbc);
#pragma unroll 1
        for (int it = i0; it <= i1; ++it) {                          // iteration `it` produces item `it` (none in the last one)
            if (it < i1) {
                const int k = (it - i0) & 1;
                if (it + 1 < i1) load_b(item_lb(it + 1), fbn);     // next item's key tile (its query tiles: below, when the query block changes)
                float* D = Dl + k * (BDB / 4);
                const int lcol = 32 * wv + li;
#pragma unroll
                for (int pt = 0; pt < 6; ++pt) {
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const i32x4 a = fa[pt][j], b = fbc[j];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.x), __int_as_float(b.x), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.y), __int_as_float(b.y), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.z), __int_as_float(b.z), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(a.w), __int_as_float(b.w), acc, 0, 0, 0);
                    }
                    if (lcol < BLN) {
                        // row 32 pt + (r & 3) + 8 (r >> 2) + 4 lh: per-lane base + compile-time offset (ds_write immediates)
                        float* dcol = D + (4 * lh) * BDP + lcol;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            constexpr int kLast = BPN - 1;
                            const int rr = 32 * pt + (r & 3) + 8 * (r >> 2);           // + 4 lh
                            if (rr + 4 <= kLast) dcol[rr * BDP] = acc[r];
                            else if (rr <= kLast && lh == 0) dcol[rr * BDP] = acc[r];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) fbc[j] = fbn[j];
                if (it + 1 < i1 && item_pb(it + 1) != cur_pb) {      // wave-uniform: the next item starts a new query block
                    cur_pb = item_pb(it + 1);
                    load_a(cur_pb);
                }
            }
            __syncthreads();
        }
    } else {
        // ---- consumers.  Thread -> key pixel l = lane of the 4 x 16 block, query pixels 32 (wave - 4) .. + 31 of the 8 x 16 block
        const int lyy = lane >> 4, lxx = lane & 15;
        int lcol9[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) lcol9[3 * a + b] = (lyy + a) * BHW + lxx + b;     // halo index of l + (a - 1, b - 1)
        const int q0 = 32 * (wv - 4);
#pragma unroll 1
        for (int it = i0; it <= i1; ++it) {                          // iteration `it` consumes item `it - 1`
            if (it > i0) {
                const int ci = it - 1, k = (ci - i0) & 1;
                const int pb = item_pb(ci), lb = item_lb(ci);
                const int py0 = (pb / p.pbx) * BSH, px0 = (pb % p.pbx) * BSW;
                const int ly = (lb / p.lbx) * BLH + lyy, lx = (lb % p.lbx) * BLW + lxx;
                const bool l_ok = ly < p.Hl && lx < p.Wl;
                const float sc = l_ok ? p.nrm[(size_t)ly * p.Wl + lx] : 0.f;
                // the query pixel of output q is wave-uniform and q is a compile-time index: every LDS address is a per-lane base
                // (this lane's nine halo columns, on this wave's first halo row) + an immediate
                const float* D = Dl + k * (BDB / 4) + (q0 >> 4) * BHW * BDP;
                // stores through a buffer descriptor: lane validity (key pixel inside the map) and the wave-uniform row validity fold into
                // the per-lane offset (out of range = dropped), the output row into the SCALAR offset -- no branch per output, so the
                // 288 LDS reads of the item's 32 outputs can be batched ahead of their sums (behind a branch each, every output waited
                // for its own nine reads)
                const unsigned lane_off = l_ok ? (unsigned)((size_t)ly * p.Wl + lx) * 4u : kOobB;
                const int qy0 = py0 + (q0 >> 4);
                // (the descriptor is based at this wave's first output row: S as a whole may exceed the 4 GiB a descriptor spans)
                const size_t s_base = ((size_t)qy0 * p.Wp + px0) * p.lds_;                      // floats, wave-uniform
                const size_t s_left = p.s_floats > s_base ? (p.s_floats - s_base) * 4 : 0;      // bytes up to the end of S
                const size_t s_span = ((size_t)p.Wp + BSW) * p.lds_ * 4;                        // two output rows of this block
                const __amdgpu_buffer_rsrc_t rs_s =
                    __builtin_amdgcn_make_buffer_rsrc(p.S + s_base, 0, (unsigned)(s_left < s_span ? s_left : s_span), 0x00020000);
                const float* dk[9];
#pragma unroll
                for (int e = 0; e < 9; ++e) dk[e] = D + lcol9[e];
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    const int pyy = q >> 4, pxx = q & 15;                    // relative to this wave's two query rows
                    float s = dk[0][(pyy * BHW + pxx) * BDP];
#pragma unroll
                    for (int e = 1; e < 9; ++e) s += dk[e][((pyy + e / 3) * BHW + pxx + e % 3) * BDP];     // same order as before: (a, b) row-major
                    const bool row_ok = qy0 + pyy < p.Hp && px0 + pxx < p.Wp;           // wave-uniform
                    const unsigned soff = (unsigned)(((size_t)pyy * p.Wp + pxx) * p.lds_ * 4);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(s * sc), rs_s, (int)(row_ok ? lane_off : kOobB), (int)(row_ok ? soff : 0u), 0);
                }
            }
            __syncthreads();
        }
    }
}

bool csa_scores_box_ok(int Ch, int ldm, int ldr) { return Ch == 32 && (ldm & 3) == 0 && (ldr & 3) == 0; }

// S[p][l] (row stride ld_s) = alpha <3x3 patch of M at p, 3x3 patch of R at l> / max(|patch of R at l|, floor); nrm: Hl*Wl floats scratch
int csa_scores_box_f32(const float* M, int ldm, int Hp, int Wp, const float* R, int ldr, int Hl, int Wl, int Ch, float alpha, float floor_,
                       float* nrm, float* S, int ld_s, hipStream_t s) {
    CIAOSR_CHECK_ARG(M && R && nrm && S && csa_scores_box_ok(Ch, ldm, ldr) && aligned16(M) && aligned16(R));
    const size_t mb = (size_t)Hp * Wp * ldm * 4, rb = (size_t)Hl * Wl * ldr * 4;
    const size_t s_floats = ((size_t)Hp * Wp - 1) * ld_s + (size_t)Hl * Wl;
    CIAOSR_CHECK_ARG(mb < 0xFFFFFF00ull && rb < 0xFFFFFF00ull && ((size_t)Wp + BSW) * ld_s * 4 < 0xFFFFFF00ull);
    {
        ProfScope prof("csa_key_norms", s);
        hipLaunchKernelGGL(csa_key_norms_kernel, dim3(ceil_div((long)Hl * Wl, 4)), dim3(256), 0, s, R, ldr, Hl, Wl, Ch, floor_, alpha, nrm);
    }
    int rc = launch_status("csa_key_norms");
    if (rc != CIAOSR_OK) return rc;
    BoxP p;
    p.M = M; p.ldm = ldm; p.m_bytes = (unsigned)mb; p.Hp = Hp; p.Wp = Wp;
    p.R = R; p.ldr = ldr; p.r_bytes = (unsigned)rb; p.Hl = Hl; p.Wl = Wl;
    p.nrm = nrm; p.S = S; p.lds_ = ld_s; p.s_floats = s_floats;
    p.pbx = ceil_div(Wp, BSW); p.lbx = ceil_div(Wl, BLW);
    p.n_lb = ceil_div(Hl, BLH) * p.lbx;
    const long n_items = (long)ceil_div(Hp, BSH) * p.pbx * p.n_lb;
    CIAOSR_CHECK_ARG(n_items < 0x7FFFFFFF);
    p.n_items = (int)n_items;
    CIAOSR_BIG_LDS(csa_scores_box_f32_kernel, kBoxLds);
    ProfScope prof("csa_scores", s);
    hipLaunchKernelGGL(csa_scores_box_f32_kernel, dim3(p.n_items < 256 ? p.n_items : 256), dim3(512), kBoxLds, s, p);
    return launch_status("csa_scores_box_f32");
}

}  // namespace ciaosr
