// LocalImplicitSRNet.forward minus the encoder (ciaosr_net.py:88-248): one C entry point that
// runs unfold -> cs_attn -> layer-1 tables -> [per work-chunk: rows, MLP chains, local attention,
// decode] on one stream.  Staged form: every stage is its own kernel with a clean roofline.
#include "ops.h"

namespace ciaosr {

struct HeadPlan {
    int H, W, C, Cn, D, Dv, J, HW;
    int wk0, wv0;            // layer-0 widths
    int wmax;                // widest hidden activation among k / v / q chains
    int qc;                  // queries per work-chunk of the STAGED route (its [4 qc][..] intermediates are the big buffers)
    int qcf;                 // queries per launch pair of the FUSED route: only Z [qcf][Dv] scales with it, so a whole 192x192
                             // tile (589 824 queries) is ONE launch pair instead of nine -- every launch ends with a drain phase
                             // in which the 4-workgroups-per-CU overlap that the kernels live on decays (probe: a workgroup lives
                             // 685k cycles, a 65 536-query launch 3.0M, so ~1/4 of every launch ran under-occupied)
    size_t csa_bytes;
};

constexpr int kLdG = 260;        // logit-table row: 256 coefficients + the bias term + pad
constexpr int kQkChunk = 65536;  // rows of (q*key) products materialised at a time

static int n_samples(int local_size) { return local_size == 1 ? 1 : (local_size == 2 ? 4 : 9); }

static HeadPlan head_plan(int H, int W, const ciaosr_head_weights_t* w, int Q) {
    HeadPlan p;
    p.H = H; p.W = W; p.C = w->channels; p.Cn = w->nonlocal_channels;
    p.D = (w->no_unfold ? 1 : 9) * p.C; p.Dv = p.D + p.Cn; p.J = n_samples(w->local_size); p.HW = H * W;
    p.wk0 = w->k.width[0];
    p.wv0 = w->v.width[0];
    int wm = 4;
    for (int i = 0; i + 1 < w->k.n_layers; ++i) wm = wm > w->k.width[i] ? wm : w->k.width[i];
    for (int i = 0; i + 1 < w->v.n_layers; ++i) wm = wm > w->v.width[i] ? wm : w->v.width[i];
    for (int i = 0; i + 1 < w->q.n_layers; ++i) wm = wm > w->q.width[i] ? wm : w->q.width[i];
    p.wmax = wm;
    p.qc = Q < 65536 ? Q : 65536;
    p.qcf = Q < (1 << 20) ? Q : (1 << 20);
    const long zcap = (long)(0xE0000000ull / ((size_t)p.Dv * sizeof(float)));      // Z is addressed through a 32-bit buffer descriptor
    if (p.qcf > zcap) p.qcf = (int)zcap;
    p.csa_bytes = p.Cn > 0 ? ciaosr_cs_attn_workspace_bytes_scale(H, W, p.C, w->nonlocal_max_scale ? w->nonlocal_max_scale : 2) : 0;
    return p;
}

static size_t head_ws_bytes(const HeadPlan& p) {
    const size_t R = (size_t)p.qc * p.J;
    size_t n = 0;
    n += (size_t)p.HW * p.Dv;                 // U
    n += (size_t)p.HW * (p.wk0 + p.wv0);      // tables
    n += 2 * R * p.wmax;                      // ping-pong activations
    n += R * p.wv0;                           // Hv (layer-1 rows of the value chain, kept while the key chain runs)
    n += R * p.D + R * p.Dv;                  // WK, WV
    n += (size_t)p.qcf * p.Dv;                // Z (sized for the fused route's chunk)
    n += (size_t)p.qc + R;                    // q_idx, k_idx (ints, same size as float)
    n += (size_t)p.HW * 9 * kLdG + (size_t)kQkChunk * p.D;   // logit table + one chunk of its GEMM rows
    n += (size_t)128 * p.D + 64;                             // bf16 mode: transposed bf16 copy of imnet_k's output layer
    n += 64;                                                 // 16-bit chained kernel: its "redo with the 128-row kernel" flag
    return n * sizeof(float) + p.csa_bytes + 32 * 256;
}

static int mlp_act(const ciaosr_mlp_t& m) { return m.act == CIAOSR_ACT_SIN || m.act == CIAOSR_ACT_COS ? m.act : CIAOSR_ACT_RELU; }

static bool mlp_ok(const ciaosr_mlp_t& m) {
    if (m.n_layers < 1 || m.n_layers > CIAOSR_MAX_LAYERS) return false;
    for (int i = 0; i < m.n_layers; ++i)
        if (!m.weight[i] || !m.bias[i] || (m.ld[i] & 3) != 0 || m.width[i] <= 0) return false;
    return true;
}

// layers 1 .. n-1 of an MLP on rows already holding the layer-0 activations.
// Returns the buffer holding the result; the last layer writes to `last_out` (ld_last) with no ReLU.
static int run_tail(const ciaosr_mlp_t& m, const float* h0, int ld0, float* bufA, float* bufB, float* last_out,
                    int ld_last, long rows, hipStream_t s, const char* tag_hidden, const char* tag_out) {
    const float* cur = h0;
    int ld_cur = ld0;
    float* pp[2] = {bufA, bufB};
    int flip = 0;
    for (int i = 1; i < m.n_layers; ++i) {
        const bool last = (i == m.n_layers - 1);
        float* dst = last ? last_out : pp[flip];
        const int ldd = last ? ld_last : m.width[i];
        int rc = gemm_f32(cur, ld_cur, m.weight[i], m.ld[i], false, dst, ldd, m.bias[i], (int)rows, m.width[i],
                          m.width[i - 1], 1.f, last ? CIAOSR_ACT_NONE : mlp_act(m), 0.f, s,
                          last ? tag_out : tag_hidden);
        if (rc != CIAOSR_OK) return rc;
        cur = dst;
        ld_cur = ldd;
        flip ^= 1;
    }
    return CIAOSR_OK;
}

// fused kernels: hidden width 256 everywhere, fragments packed, 4 key samples
static bool chain_fused_ok(const ciaosr_mlp_t& m, bool is_q, bool bf16) {
    if (m.n_layers < 2 || mlp_act(m) != CIAOSR_ACT_RELU) return false;
    for (int i = 0; i + 1 < m.n_layers; ++i)
        if (m.width[i] != 256) return false;
    for (int i = is_q ? 0 : 1; i < m.n_layers - (is_q ? 1 : 0); ++i)
        if (!(bf16 ? m.frag16[i] : (const void*)m.frag[i])) return false;
    return true;
}

static void fill_chain(FusedChain& c, const ciaosr_mlp_t& m, const float* table, int fan, bool bf16, bool lo) {
    c.table = table;
    c.tail = m.weight[0] + fan;
    c.ld_tail = m.ld[0];
    c.n_hidden = m.n_layers - 2;
    for (int i = 0; i < c.n_hidden; ++i) {
        c.frag_hidden[i] = bf16 ? m.frag16[i + 1] : (const void*)m.frag[i + 1];
        c.frag_hidden_lo[i] = (bf16 && lo) ? m.frag16_lo[i + 1] : nullptr;
        c.bias_hidden[i] = m.bias[i + 1];
    }
    c.frag_out = bf16 ? m.frag16[m.n_layers - 1] : (const void*)m.frag[m.n_layers - 1];
    c.frag_out_lo = (bf16 && lo) ? m.frag16_lo[m.n_layers - 1] : nullptr;
    c.bias_out = m.bias[m.n_layers - 1];
    c.n_out = m.width[m.n_layers - 1];
}

}  // namespace ciaosr

using namespace ciaosr;

extern "C" size_t ciaosr_head_workspace_bytes(int H, int W, const ciaosr_head_weights_t* w, int Q) {
    if (!w || H <= 0 || W <= 0 || Q <= 0) return 0;
    return head_ws_bytes(head_plan(H, W, w, Q));
}

static int head_forward(const float* feat_hwc, int H, int W, const ciaosr_head_weights_t* w,
                        const ciaosr_csattn_weights_t* csattn, const float* x_lr_nchw, const float* coord,
                        const float* cell, int Q, int chunk, float* rgb, const ciaosr_options_t* opt, void* workspace,
                        size_t workspace_bytes, void* stream_, Prec prec) {
    const bool bf16 = prec != kF32;                  // a 16-bit MFMA mode (bf16 or f16 entry)
    const int route = opt ? opt->head_route : 0;
    // hi + lo weight pairs: the bf16 entry unless single is asked for, the f16 entry when pairs are asked for
    const bool lo = (prec == kBF16 && (!(opt && opt->bf16_single) || (opt && opt->f16_pairs == 2))) || (prec == kF16 && opt && opt->f16_pairs);
    // f16_pairs = 2 / 3 ("f16x3" / "f16x3-fast"): the activations of the three MLP chains as half pairs too (head_fused_wide.hip), every table in fp32
    // (round 6: the _bf16 entry takes f16_pairs = 2 too -- "bf16x3": bf16 hi + lo weights AND activations; bf16_single is ignored then)
    const bool x3 = (prec == kF16 && opt && opt->f16_pairs >= 2) || (prec == kBF16 && opt && opt->f16_pairs == 2);
    // f16x3 runs the wide-workgroup kernels (head_fused_wide.hip: its two activation arrays leave room for one workgroup per CU);
    // f16 / f16-pairs keep the 128-row kernels with two workgroups per CU (head_fused_h16.hip) and take the wide form -- 256 rows, half
    // the weight stream per MFMA, measured equal in time: one workgroup per CU exposes its gather phases -- only with head_route bit 3
    const bool wide16 = x3 || (prec == kF16 && (route & CIAOSR_HEAD_WIDE_WG));
    const int wide_mode = x3 ? 2 : (lo ? 1 : 0);
    CIAOSR_CHECK_ARG(feat_hwc && w && coord && cell && rgb && workspace && H >= 1 && W >= 1 && Q >= 1);
    CIAOSR_CHECK_ARG(options_ok(opt));
    CIAOSR_CHECK_ARG(w->channels >= 4 && (w->channels & 3) == 0 && (w->nonlocal_channels & 3) == 0);
    CIAOSR_CHECK_ARG(w->local_size >= 1 && w->local_size <= 3 && w->softmax_scale != 0.f);
    CIAOSR_CHECK_ARG(mlp_ok(w->q) && mlp_ok(w->k) && mlp_ok(w->v));
    CIAOSR_CHECK_ARG((w->nonlocal_channels > 0) == (csattn != nullptr));
    hipStream_t s = (hipStream_t)stream_;
    const HeadPlan p = head_plan(H, W, w, Q);
    // dims wiring of LocalImplicitSRNet.__init__ (ciaosr_net.py:61-76)
    CIAOSR_CHECK_ARG(w->k.in_dim == p.D + 4 && w->k.width[w->k.n_layers - 1] == p.D);
    CIAOSR_CHECK_ARG(w->v.in_dim == p.Dv + 4 && w->v.width[w->v.n_layers - 1] == p.Dv);
    CIAOSR_CHECK_ARG(w->q.in_dim == p.Dv && w->q.width[w->q.n_layers - 1] == 3);
    CIAOSR_CHECK_ARG(w->k.n_layers >= 2 && w->v.n_layers >= 2 && w->q.n_layers >= 2);
    CIAOSR_CHECK_ARG((p.wk0 & 3) == 0 && (p.wv0 & 3) == 0);
    const int n_scales = csattn ? p.Cn / p.C : 0;            // csattn = host array, one struct per entry of multi_scale
    if (csattn) {
        CIAOSR_CHECK_ARG(p.Cn == n_scales * p.C && n_scales >= 1 && n_scales <= 3);
        for (int i = 0; i < n_scales; ++i) {
            const int sc = csattn[i].scale ? csattn[i].scale : 2;
            CIAOSR_CHECK_ARG(csattn[i].channels == p.C && sc <= (w->nonlocal_max_scale ? w->nonlocal_max_scale : 2));
        }
    }
    if (workspace_bytes < head_ws_bytes(p)) return CIAOSR_ERR_WORKSPACE;
    CIAOSR_CHECK_ARG((size_t)p.HW * p.Dv * sizeof(float) < 0xFFFFFF00ull);   // 32-bit buffer offsets into U

    Arena ar(workspace, workspace_bytes);
    const size_t R = (size_t)p.qc * p.J;
    float* U = ar.take<float>((size_t)p.HW * p.Dv);
    float* Tk = ar.take<float>((size_t)p.HW * p.wk0);
    float* Tv = ar.take<float>((size_t)p.HW * p.wv0);
    float* bufA = ar.take<float>(R * p.wmax);
    float* bufB = ar.take<float>(R * p.wmax);
    float* Hv = ar.take<float>(R * p.wv0);
    float* WK = ar.take<float>(R * p.D);
    float* WV = ar.take<float>(R * p.Dv);
    float* Z = ar.take<float>((size_t)p.qcf * p.Dv);
    int* q_idx = ar.take<int>(p.qc);
    int* k_idx = ar.take<int>(R);
    float* G = ar.take<float>((size_t)p.HW * 9 * kLdG);
    float* QK = ar.take<float>((size_t)kQkChunk * p.D);
    unsigned short* W5T = reinterpret_cast<unsigned short*>(ar.take<float>((size_t)128 * p.D + 64));
    int* chain_flag = ar.take<int>(64);
    char* csa_ws = ar.take<char>(p.csa_bytes);
    if (!ar.ok) return CIAOSR_ERR_WORKSPACE;

    int rc;
#define RUN(x) do { rc = (x); if (rc != CIAOSR_OK) return rc; } while (0)
    // unfold rows U[:, :9C] (net:132-136) and the non-local map into U[:, 9C:] (net:134-137)
    if (w->no_unfold)     // feat_unfold=False (net:139-141): the "unfold" row is the pixel's C features (a 1x1 patch)
        RUN(patch_rows(feat_hwc, p.C, H, W, p.C, 1, 1, 0, H, W, U, p.Dv, 0, 0.f, s, "head_unfold"));
    else
        RUN(patch_rows(feat_hwc, p.C, H, W, p.C, 3, 1, 1, H, W, U, p.Dv, 0, 0.f, s, "head_unfold"));
    for (int i = 0; i < n_scales; ++i)     // one C-column slice per scale, in the order of multi_scale (csa:528 torch.cat(res_y, dim=1))
        RUN((prec == kF16 ? ciaosr_cs_attn_f16 : prec == kBF16 ? ciaosr_cs_attn_bf16 : ciaosr_cs_attn_f32)(feat_hwc, p.C, H, W, csattn + i, U + p.D + (size_t)i * p.C, p.Dv, opt, csa_ws,
                                                              p.csa_bytes, stream_));
    const bool fused = !(route & CIAOSR_HEAD_STAGED) && w->local_size == 2 && chain_fused_ok(w->k, false, bf16) &&
                       chain_fused_ok(w->v, false, bf16) && chain_fused_ok(w->q, true, bf16) && (p.Dv & 7) == 0;
    if (bf16 && !fused) return CIAOSR_ERR_UNSUPPORTED;   // the 16-bit modes exist for the fused kernels only
    if (x3) {                                            // the pair kernels read every lo fragment unconditionally
        for (int i = 1; i < w->k.n_layers; ++i) CIAOSR_CHECK_ARG(w->k.frag16_lo[i]);
        for (int i = 1; i < w->v.n_layers; ++i) CIAOSR_CHECK_ARG(w->v.frag16_lo[i]);
        for (int i = 0; i + 1 < w->q.n_layers; ++i) CIAOSR_CHECK_ARG(w->q.frag16_lo[i]);
    }
    // exact layer-1 hoist: T = U . W1[:, :fan]^T + b1, one row per LR pixel
    // f16 mode: on the 16-bit GEMM from a half copy of U (the staged route's activation buffers are free on the fused route)
    const size_t u16_bytes = (size_t)p.HW * p.Dv * 2 + 256, w16_bytes = (size_t)(p.wk0 + p.wv0) * p.Dv * 2 + 512;
    if (prec == kF16 && !lo && (p.D & 7) == 0 && (p.Dv & 7) == 0 && u16_bytes + w16_bytes <= R * p.wmax * sizeof(float) &&
        (size_t)p.HW * p.Dv * 2 < 0xFFFFFF00ull) {
        const H16Ops& h = h16_ops(prec);
        unsigned short* U16 = reinterpret_cast<unsigned short*>(bufA);
        unsigned short* Wk16 = U16 + round_up((size_t)p.HW * p.Dv, 128);
        unsigned short* Wv16 = Wk16 + round_up((size_t)p.wk0 * p.D, 128);
        RUN(h.cast_rows(U, p.Dv, U16, p.Dv, p.HW, p.Dv, s));
        RUN(h.cast_rows(w->k.weight[0], w->k.ld[0], Wk16, p.D, p.wk0, p.D, s));
        RUN(h.cast_rows(w->v.weight[0], w->v.ld[0], Wv16, p.Dv, p.wv0, p.Dv, s));
        RUN(h.conv1x1(U16, p.Dv, Wk16, p.D, w->k.bias[0], nullptr, 0, Tk, p.wk0, nullptr, 0, nullptr, 0, p.HW, p.wk0, p.D, s, "head_table_f16"));
        RUN(h.conv1x1(U16, p.Dv, Wv16, p.Dv, w->v.bias[0], nullptr, 0, Tv, p.wv0, nullptr, 0, nullptr, 0, p.HW, p.wv0, p.Dv, s, "head_table_f16"));
    } else if (gemm_small_ok(p.HW, p.wk0, p.D, p.Dv, w->k.ld[0]) && gemm_small_ok(p.HW, p.wv0, p.Dv, p.Dv, w->v.ld[0]) && p.HW <= 4096) {
        RUN(gemm_small_f32(U, p.Dv, w->k.weight[0], w->k.ld[0], w->k.bias[0], Tk, p.wk0, nullptr, 0, nullptr, 0, p.HW, p.wk0, p.D,
                           CIAOSR_ACT_NONE, 0.f, 1.f, s, "head_table"));
        RUN(gemm_small_f32(U, p.Dv, w->v.weight[0], w->v.ld[0], w->v.bias[0], Tv, p.wv0, nullptr, 0, nullptr, 0, p.HW, p.wv0, p.Dv,
                           CIAOSR_ACT_NONE, 0.f, 1.f, s, "head_table"));
    } else {
        RUN(gemm_f32(U, p.Dv, w->k.weight[0], w->k.ld[0], false, Tk, p.wk0, w->k.bias[0], p.HW, p.wk0, p.D, 1.f,
                     CIAOSR_ACT_NONE, 0.f, s, "head_table"));
        RUN(gemm_f32(U, p.Dv, w->v.weight[0], w->v.ld[0], false, Tv, p.wv0, w->v.bias[0], p.HW, p.wv0, p.Dv, 1.f,
                     CIAOSR_ACT_NONE, 0.f, s, "head_table"));
    }

    // logit table of imnet_k's output layer (exact fold, head_ops.hip): pays off when queries outnumber LR pixels
    const bool use_table = fused && !(route & CIAOSR_HEAD_NO_LOGIT_TABLE) && w->k.width[w->k.n_layers - 1] == p.D && w->k.width[w->k.n_layers - 2] == 256 &&
                           (long)Q * p.J > (long)p.HW * 9 && (size_t)p.HW * 9 * kLdG * sizeof(float) < 0xFFFFFF00ull;
    if (use_table) {
        const int last = w->k.n_layers - 1;
        const long total = (long)p.HW * 9;
        // 16-bit modes: the table GEMM on the 16-bit MFMA (fp32 table out).  (Half-pairs mode: the exact-fp32 GEMM here was measured
        // and changes nothing -- max |delta| 1.04e-3 -> 1.14e-3 on the full-tile vector, +0.9 ms: W5's rounding is not what limits it.)
        const bool table16 = bf16 && !x3 && (p.D & 7) == 0;
        if (table16) RUN(transpose_cast_h16(w->k.weight[last], w->k.ld[last], p.D, 256, W5T, prec == kF16, s));
        // fp32, C = 64: nine Winograd convolutions of the product maps Pi_o = F . shift_o(F) (same sums as the GEMM rows below,
        // re-associated through the transform; 2.25x fewer multiplies).  The maps live where the GEMM would keep its row chunk.
        const bool table_wino = (prec == kF32 || x3) && w->k_out_wino && !(route & CIAOSR_HEAD_TABLE_GEMM) && p.C == 64 && !w->no_unfold &&
                                p.HW >= 512 && p.HW <= kQkChunk;
        if (table_wino) {
            RUN(qk_maps(feat_hwc, p.C, p.C, H, W, QK, s));
            if (w->k_out_wino4 && !(route & CIAOSR_HEAD_TABLE_WINO2)) RUN(wino4_table_f32(QK, H, W, w->k_out_wino4, 4, G, kLdG, s));
            else RUN(wino_table_f32(QK, H, W, w->k_out_wino, 4, G, kLdG, s));
            RUN(qk_rows(U, p.Dv, p.D, H, W, 0, (int)total, w->k.bias[last], nullptr, G, kLdG, 3, s));
        }
        for (long r0 = 0; r0 < total && !table_wino; r0 += kQkChunk) {
            const int nr = (int)((total - r0) < kQkChunk ? (total - r0) : kQkChunk);
            RUN(qk_rows(U, p.Dv, p.D, H, W, r0, nr, w->k.bias[last], QK, G, kLdG, table16 ? (int)prec : 0, s));
            if (table16) {
                RUN(h16_ops(prec).gemm_nt(reinterpret_cast<const unsigned short*>(QK), p.D, W5T, p.D, G + (size_t)r0 * kLdG, kLdG, false, nr,
                                          256, p.D, 1.f, s, prec == kF16 ? "head_logit_table_f16" : "head_logit_table_bf16"));
                continue;
            }
            // G[r][n] = sum_d QK[r][d] * W5k[d][n]: the Linear weight [D][256] is the [K][N] operand as stored
            RUN(gemm_f32(QK, p.D, w->k.weight[last], w->k.ld[last], true, G + (size_t)r0 * kLdG, kLdG, nullptr, nr, 256, p.D,
                         1.f, CIAOSR_ACT_NONE, 0.f, s, "head_logit_table"));
        }
    }
    const int step = fused ? p.qcf : p.qc;
    for (long q0 = 0; q0 < Q; q0 += step) {
        const int nq = (int)((Q - q0) < step ? (Q - q0) : step);
        const long rows = (long)nq * p.J;
        if (fused) {
            FusedKVP kp;
            kp.coord = coord; kp.cell = cell; kp.q0 = q0; kp.nq = nq; kp.chunk = chunk; kp.H = H; kp.W = W;
            kp.U = U; kp.ldu = p.Dv; kp.D = p.D; kp.Dv = p.Dv;
            kp.u_bytes = (unsigned)((size_t)p.HW * p.Dv * sizeof(float));
            fill_chain(kp.k, w->k, Tk, p.D, bf16, lo);
            fill_chain(kp.v, w->v, Tv, p.Dv, bf16, lo);
            kp.softmax_scale = w->softmax_scale;
            kp.Z = Z; kp.ldz = p.Dv;
            kp.rows_per_wg = opt ? opt->kv_rows : 0;
            kp.G = use_table ? G : nullptr; kp.ldg = kLdG; kp.g_bytes = (unsigned)((size_t)p.HW * 9 * kLdG * sizeof(float));
            kp.gate = nullptr;
            // 16-bit default: the weights-stationary, register-chained kernel (head_chain_h16.hip) where its weight stream is given and the
            // logit table exists; the 128-row kernel is launched behind it, gated on the flag the chained kernel raises when a key leaves its
            // query's 3x3 neighbourhood (cannot happen for 0 < cell < 1): it then redoes the launch, else it returns at once
            const void* blob = lo ? w->chain16_pairs : w->chain16;
            const bool chained = bf16 && !wide16 && !x3 && use_table && blob && !(route & CIAOSR_HEAD_NO_CHAIN) && h16_ops(prec).head_chain_ok(w) &&
                                 (size_t)nq * p.Dv * 2 < 0xFFFFFF00ull;
            if (chained) {
                if (hipMemsetAsync(chain_flag, 0, sizeof(int), s) != hipSuccess) return CIAOSR_ERR_LAUNCH;
                RUN(h16_ops(prec).head_kv_chain(kp, w, blob, lo ? 1 : 0, opt ? opt->query_grid_w : 0, chain_flag, s));
                kp.gate = chain_flag;
            }
            RUN(wide16 ? (prec == kF16 ? f16::wide::head_kv_fused_wide(kp, wide_mode, s) : b16::wide::head_kv_fused_wide(kp, wide_mode, s))
                       : bf16 ? h16_ops(prec).head_kv_fused(kp, s) : head_kv_fused(kp, s));
            const ciaosr_mlp_t& mq = w->q;
            FusedQP qp;
            qp.Z = Z; qp.ldz = p.Dv; qp.Dv = p.Dv;
            qp.frag_in = bf16 ? mq.frag16[0] : (const void*)mq.frag[0]; qp.bias_in = mq.bias[0];
            qp.frag_in_lo = (bf16 && lo) ? mq.frag16_lo[0] : nullptr;
            qp.nj_in = bf16 ? (p.Dv + 15) / 16 : (p.Dv + 7) / 8;
            qp.n_hidden = mq.n_layers - 2;
            for (int i = 0; i < qp.n_hidden; ++i) {
                qp.frag_hidden[i] = bf16 ? mq.frag16[i + 1] : (const void*)mq.frag[i + 1];
                qp.frag_hidden_lo[i] = (bf16 && lo) ? mq.frag16_lo[i + 1] : nullptr;
                qp.bias_hidden[i] = mq.bias[i + 1];
            }
            qp.w_last = mq.weight[mq.n_layers - 1]; qp.ld_last = mq.ld[mq.n_layers - 1];
            qp.b_last = mq.bias[mq.n_layers - 1];
            qp.rows_per_wg = opt ? opt->decode_rows : 0;
            qp.x_lr = x_lr_nchw; qp.coord = coord; qp.q0 = q0; qp.nq = nq; qp.H = H; qp.W = W; qp.rgb = rgb;
            // imnet_q through the same weights-stationary form where the blob carries its stream (Dv a multiple of 128, 256-wide layers)
            if (chained && !(route & CIAOSR_HEAD_NO_DECODE_CHAIN) && h16_ops(prec).head_decode_chain_ok(w)) {
                const unsigned char* qblob = reinterpret_cast<const unsigned char*>(blob) + h16_ops(prec).head_kv_chain_bytes(w, lo ? 1 : 0);
                RUN(h16_ops(prec).head_decode_chain(qp, w, qblob, lo ? 1 : 0, s));
                continue;
            }
            RUN(wide16 ? (prec == kF16 ? f16::wide::head_decode_fused_wide(qp, wide_mode, s) : b16::wide::head_decode_fused_wide(qp, wide_mode, s))
                       : bf16 ? h16_ops(prec).head_decode_fused(qp, s) : head_decode_fused(qp, s));
            continue;
        }
        HeadRowsP hp;
        hp.coord = coord; hp.cell = cell; hp.q0 = q0; hp.nq = nq; hp.chunk = chunk; hp.H = H; hp.W = W;
        hp.local_size = w->local_size; hp.J = p.J;
        hp.Tk = Tk; hp.Tv = Tv;
        hp.tailK = w->k.weight[0] + p.D;      // columns [9C, 9C+4) of layer 0, stride ld -> packed copy below
        hp.tailV = w->v.weight[0] + p.Dv;
        hp.wk0 = p.wk0; hp.wv0 = p.wv0; hp.relu_k = mlp_act(w->k); hp.relu_v = mlp_act(w->v);
        hp.Hk = bufA; hp.Hv = Hv; hp.q_idx = q_idx; hp.k_idx = k_idx;
        hp.ld_tail_k = w->k.ld[0]; hp.ld_tail_v = w->v.ld[0];
        RUN(head_rows(hp, s));
        // imnet_k layers 1..n-1 -> WK   (bufA holds layer-0 rows; ping-pong through bufB/bufA)
        RUN(run_tail(w->k, bufA, p.wk0, bufB, bufA, WK, p.D, rows, s, "mlp_hidden", "mlp_out_k"));
        // imnet_v: layer-0 rows are in Hv
        RUN(run_tail(w->v, Hv, p.wv0, bufA, bufB, WV, p.Dv, rows, s, "mlp_hidden", "mlp_out_v"));
        LocalAttnP lp{U, p.Dv, p.D, p.Dv, q_idx, k_idx, WK, p.D, WV, p.Dv, Z, p.Dv, nq, p.J, w->softmax_scale};
        RUN(local_attention(lp, s));
        // imnet_q: layer 0 on Z, hidden layers, last layer fused with the bilinear residual
        const ciaosr_mlp_t& mq = w->q;
        const float* cur = Z;
        int ld_cur = p.Dv, k_cur = p.Dv;
        float* pp[2] = {bufA, bufB};
        int flip = 0;
        for (int i = 0; i + 1 < mq.n_layers; ++i) {
            RUN(gemm_f32(cur, ld_cur, mq.weight[i], mq.ld[i], false, pp[flip], mq.width[i], mq.bias[i], nq,
                         mq.width[i], k_cur, 1.f, mlp_act(mq), 0.f, s, i == 0 ? "mlp_in_q" : "mlp_hidden_q"));
            cur = pp[flip]; ld_cur = mq.width[i]; k_cur = mq.width[i];
            flip ^= 1;
        }
        DecodeP dp{cur, ld_cur, k_cur, mq.weight[mq.n_layers - 1], mq.ld[mq.n_layers - 1],
                   mq.bias[mq.n_layers - 1], x_lr_nchw, coord, q0, nq, H, W, rgb};
        RUN(decode_residual(dp, s));
    }
#undef RUN
    return CIAOSR_OK;
}

extern "C" int ciaosr_head_forward_f32(const float* feat_hwc, int H, int W, const ciaosr_head_weights_t* w,
                                       const ciaosr_csattn_weights_t* csattn, const float* x_lr_nchw,
                                       const float* coord, const float* cell, int Q, int chunk, float* rgb,
                                       const ciaosr_options_t* opt, void* workspace, size_t workspace_bytes, void* stream) {
    return head_forward(feat_hwc, H, W, w, csattn, x_lr_nchw, coord, cell, Q, chunk, rgb, opt, workspace, workspace_bytes,
                        stream, kF32);
}

extern "C" int ciaosr_head_forward_bf16(const float* feat_hwc, int H, int W, const ciaosr_head_weights_t* w,
                                        const ciaosr_csattn_weights_t* csattn, const float* x_lr_nchw,
                                        const float* coord, const float* cell, int Q, int chunk, float* rgb,
                                        const ciaosr_options_t* opt, void* workspace, size_t workspace_bytes, void* stream) {
    return head_forward(feat_hwc, H, W, w, csattn, x_lr_nchw, coord, cell, Q, chunk, rgb, opt, workspace, workspace_bytes,
                        stream, kBF16);
}

extern "C" int ciaosr_head_forward_f16(const float* feat_hwc, int H, int W, const ciaosr_head_weights_t* w,
                                       const ciaosr_csattn_weights_t* csattn, const float* x_lr_nchw,
                                       const float* coord, const float* cell, int Q, int chunk, float* rgb,
                                       const ciaosr_options_t* opt, void* workspace, size_t workspace_bytes, void* stream) {
    return head_forward(feat_hwc, H, W, w, csattn, x_lr_nchw, coord, cell, Q, chunk, rgb, opt, workspace, workspace_bytes,
                        stream, kF16);
}

// ---- staged MLPRefiner (mlp_refiner.py:87-102), layer by layer, no hoist -------------------------------------
static int mlp_wmax(const ciaosr_mlp_t* m) {
    int w = 0;
    for (int i = 0; i + 1 < m->n_layers; ++i) w = m->width[i] > w ? m->width[i] : w;
    return (w + 3) & ~3;
}

extern "C" size_t ciaosr_mlp_workspace_bytes(const ciaosr_mlp_t* m, int rows) {
    if (!m || rows <= 0 || m->n_layers < 1 || m->n_layers > CIAOSR_MAX_LAYERS) return 0;
    return 2 * ((size_t)rows * mlp_wmax(m) * sizeof(float) + 256);
}

extern "C" int ciaosr_mlp_forward_f32(const float* x, int ld_x, const ciaosr_mlp_t* m, int n_run, int rows, float* out,
                                      int ld_out, void* workspace, size_t workspace_bytes, void* stream) {
    CIAOSR_CHECK_ARG(x && m && out && rows > 0 && mlp_ok(*m) && ld_x >= m->in_dim);
    CIAOSR_CHECK_ARG(n_run >= 0 && n_run <= m->n_layers);
    const int n = n_run ? n_run : m->n_layers;
    if (workspace_bytes < ciaosr_mlp_workspace_bytes(m, rows)) return CIAOSR_ERR_WORKSPACE;
    Arena ar(workspace, workspace_bytes);
    const int wmax = mlp_wmax(m);
    float* pp[2] = {ar.take<float>((size_t)rows * wmax), ar.take<float>((size_t)rows * wmax)};
    if (!ar.ok) return CIAOSR_ERR_WORKSPACE;
    const float* cur = x;
    int ld_cur = ld_x, k_cur = m->in_dim;
    for (int i = 0; i < n; ++i) {
        const bool last = i + 1 == n;
        float* dst = last ? out : pp[i & 1];
        const int ldd = last ? ld_out : wmax;
        int rc = gemm_f32(cur, ld_cur, m->weight[i], m->ld[i], false, dst, ldd, m->bias[i], rows, m->width[i], k_cur, 1.f,
                          i + 1 == m->n_layers ? CIAOSR_ACT_NONE : mlp_act(*m), 0.f, (hipStream_t)stream, "mlp_layer");
        if (rc != CIAOSR_OK) return rc;
        cur = dst; ld_cur = ldd; k_cur = m->width[i];
    }
    return CIAOSR_OK;
}

// ---- staged MLP with 16-bit operands (SURVEY 8(b-2) "ciaosr_mlp5_bf16"): every Linear on the 16-bit MFMA GEMM, activations 16-bit between layers,
// fp32 accumulation, biases and the last layer's output fp32.  ReLU MLPs only (what the 16-bit modes are defined for).
static size_t round8(size_t v) { return (v + 7) & ~(size_t)7; }
extern "C" size_t ciaosr_mlp_workspace_bytes_16(const ciaosr_mlp_t* m, int rows) {
    if (!m || rows <= 0 || !mlp_ok(*m)) return 0;
    size_t wmax = 0, kmax = round8((size_t)m->in_dim);
    for (int i = 0; i < m->n_layers; ++i) {
        wmax = std::max(wmax, round8((size_t)m->width[i]));
        if (i + 1 < m->n_layers) kmax = std::max(kmax, round8((size_t)m->width[i]));
    }
    // x as 16 bits, two ping-pong activation buffers, one layer's weights as 16 bits; 256 B of slack per buffer for alignment
    return ((size_t)rows * round8((size_t)m->in_dim) + 2 * (size_t)rows * wmax + wmax * kmax) * 2 + 4 * 256;
}

template <typename CastFn, typename LinFn>
static int mlp_forward_16(CastFn cast_rows, LinFn linear, const float* x, int ld_x, const ciaosr_mlp_t* m, int rows, float* out, int ld_out,
                          void* workspace, size_t workspace_bytes, hipStream_t s) {
    CIAOSR_CHECK_ARG(x && m && out && rows > 0 && mlp_ok(*m) && ld_x >= m->in_dim && (ld_x & 3) == 0 && (ld_out & 3) == 0);
    CIAOSR_CHECK_ARG(m->n_layers == 1 || mlp_act(*m) == CIAOSR_ACT_RELU);
    // cast_rows converts whole 4-column groups: with in_dim % 4 != 0 the group past in_dim would carry x padding / the next weight row
    // into the contraction (the model's own in_dim is 9C + 4 or 10C + 4 with C % 4 == 0)
    CIAOSR_CHECK_ARG((m->in_dim & 3) == 0);
    for (int i = 0; i < m->n_layers; ++i) CIAOSR_CHECK_ARG((m->width[i] & 3) == 0 || i + 1 == m->n_layers);
    if (workspace_bytes < ciaosr_mlp_workspace_bytes_16(m, rows)) return CIAOSR_ERR_WORKSPACE;
    Arena ar(workspace, workspace_bytes);
    size_t wmax = 0, kmax = round8((size_t)m->in_dim);
    for (int i = 0; i < m->n_layers; ++i) {
        wmax = std::max(wmax, round8((size_t)m->width[i]));
        if (i + 1 < m->n_layers) kmax = std::max(kmax, round8((size_t)m->width[i]));
    }
    const int k0 = (int)round8((size_t)m->in_dim);
    unsigned short* x16 = ar.take<unsigned short>((size_t)rows * k0);
    unsigned short* pp[2] = {ar.take<unsigned short>((size_t)rows * wmax), ar.take<unsigned short>((size_t)rows * wmax)};
    unsigned short* w16 = ar.take<unsigned short>(wmax * kmax);
    if (!ar.ok) return CIAOSR_ERR_WORKSPACE;
    // cast_rows zeroes the pad columns [cols, ld_dst): K is padded to a multiple of 8 on both operands
    int rc = cast_rows(x, ld_x, x16, k0, (long)rows, (m->in_dim + 3) & ~3, s);
    if (rc != CIAOSR_OK) return rc;
    const unsigned short* cur = x16;
    int ld_cur = k0, k_cur = k0, k_real = m->in_dim;
    for (int i = 0; i < m->n_layers; ++i) {
        const bool last = i + 1 == m->n_layers;
        const int N = m->width[i];
        rc = cast_rows(m->weight[i], m->ld[i], w16, k_cur, (long)N, (k_real + 3) & ~3, s);
        if (rc != CIAOSR_OK) return rc;
        if (last) {
            // an output width that is no multiple of 4 (imnet_q: 3) goes through a padded scratch row in the free ping-pong buffer
            if ((N & 3) == 0) rc = linear(cur, ld_cur, w16, k_cur, m->bias[i], false, out, ld_out, false, rows, N, k_cur, s, "mlp_layer_16");
            else return CIAOSR_ERR_UNSUPPORTED;
        } else {
            const int ldn = (int)round8((size_t)N);
            unsigned short* dst = pp[i & 1];
            if (ldn != N && hipMemsetAsync(dst, 0, (size_t)rows * ldn * 2, s) != hipSuccess) return CIAOSR_ERR_LAUNCH;
            rc = linear(cur, ld_cur, w16, k_cur, m->bias[i], true, dst, ldn, true, rows, N, k_cur, s, "mlp_layer_16");
            cur = dst; ld_cur = ldn; k_cur = ldn; k_real = N;
        }
        if (rc != CIAOSR_OK) return rc;
    }
    return CIAOSR_OK;
}

extern "C" int ciaosr_mlp_forward_bf16(const float* x, int ld_x, const ciaosr_mlp_t* m, int rows, float* out, int ld_out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return mlp_forward_16(b16::cast_rows_h16, b16::linear_h16, x, ld_x, m, rows, out, ld_out, workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int ciaosr_mlp_forward_f16(const float* x, int ld_x, const ciaosr_mlp_t* m, int rows, float* out, int ld_out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    return mlp_forward_16(f16::cast_rows_h16, f16::linear_h16, x, ld_x, m, rows, out, ld_out, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- weight stream of the weights-stationary 16-bit head (head_chain_h16.hip) ---------------------------------------------------------
static bool chain_weights_ok(const ciaosr_head_weights_t* w) {
    if (!w || !mlp_ok(w->k) || !mlp_ok(w->v)) return false;
    return b16::head_chain_ok(w);
}

extern "C" size_t ciaosr_head_chain_bytes(const ciaosr_head_weights_t* w, int pairs) {
    if (!chain_weights_ok(w)) return 0;
    return b16::head_chain_bytes(w, pairs ? 1 : 0);
}

extern "C" int ciaosr_pack_head_chain_bf16(const ciaosr_head_weights_t* w, int pairs, void* out, void* stream) {
    CIAOSR_CHECK_ARG(out && (pairs == 0 || pairs == 1));
    if (!chain_weights_ok(w)) return CIAOSR_ERR_UNSUPPORTED;
    return b16::pack_head_chain(w, pairs, out, (hipStream_t)stream);
}

extern "C" int ciaosr_pack_head_chain_f16(const ciaosr_head_weights_t* w, int pairs, void* out, void* stream) {
    CIAOSR_CHECK_ARG(out && (pairs == 0 || pairs == 1));
    if (!chain_weights_ok(w)) return CIAOSR_ERR_UNSUPPORTED;
    return f16::pack_head_chain(w, pairs, out, (hipStream_t)stream);
}
