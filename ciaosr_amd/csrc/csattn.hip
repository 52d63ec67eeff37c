// CrossScaleAttention for scale 2 (arch_csnln.py:430-532) as a sequence of launches on one stream.
//
//   xp   = reflect-pad to even H,W                                   pad_reflect          (:444-449)
//   E    = PReLU(conv1x1_assembly(xp)),  M = PReLU(conv1x1_match1(xp))   gemm (MFMA)      (:452-453)
//   R    = PReLU(conv1x1_match2(avgpool2(xp)))                       avgpool2 + gemm      (:474-475)
//   Qp   = 3x3 patches of M (zero pad 1)            [HpWp][9C/2]     patch_rows           (:498-499)
//   Kn   = L2-normalised 3x3 patches of R           [L][9C/2]        patch_rows(norm)     (:476-496)
//   V    = 6x6 stride-2 patches of E (zero pad 2)   [L][36C]         patch_rows           (:462-469)
//   S    = 10 * Qp . Kn^T                           [HpWp][L]        gemm NT (MFMA)       (:499-500)
//   P    = softmax_L(S)                                               softmax_rows         (:505)
//   O    = P . V                                     [HpWp][36C]      gemm NN (MFMA)       (:511)
//   Y    = fold(O) (gather form, stride 2, pad 2)    [2Hp][2Wp][C]    fold                 (:511)
//   out  = (conv3x3 s2 p1 (Y) + b) / 6, cropped      [H][W][C]        patch_rows + gemm    (:516-526)
//
// The score matrix is materialised in HBM (1.36 GB at the 192x192 tile: < 1 % of 288 GB and two
// passes at HBM speed, against 1.76 TFLOP of MFMA work).
#include "ops.h"

namespace ciaosr {

struct CsaPlan {
    int H, W, C, Hp, Wp, L, Lld, Ch;
    size_t n_xp, n_E, n_M, n_x2, n_R, n_Qp, n_Kn, n_V, n_S, n_O, n_Y, n_Yp;
    size_t n_PE, n_Vp, n_Ov;   // composed fold+down form
    int Lld8; size_t n_P16;    // bf16 mode: probabilities [HpWp][Lld8] bf16
};

static CsaPlan csa_plan(int H, int W, int C, int sc = 2) {
    CsaPlan p;
    p.H = H; p.W = W; p.C = C; p.Ch = (int)round_up(C / 2, 4);  // zero-padded half width
    p.Hp = (int)round_up((size_t)H, sc); p.Wp = (int)round_up((size_t)W, sc);        // mod_pad to the scale (csa:438-444)
    p.L = (p.Hp / sc) * (p.Wp / sc);
    p.Lld = (int)round_up(p.L, 4);
    const size_t HW = (size_t)p.Hp * p.Wp;
    p.n_xp = HW * C;
    p.n_E = HW * C;
    p.n_M = HW * p.Ch;
    p.n_x2 = (size_t)p.L * C;
    p.n_R = (size_t)p.L * p.Ch;
    p.n_Qp = HW * 9 * p.Ch;
    p.n_Kn = (size_t)p.L * 9 * p.Ch;
    p.n_V = (size_t)p.L * 9 * sc * sc * C;          // (3s)x(3s) patches
    p.n_S = HW * p.Lld;
    p.n_O = HW * 9 * sc * sc * C;
    p.n_Y = (size_t)sc * sc * HW * C;
    p.n_Yp = (size_t)H * W * 9 * C;
    p.n_PE = (size_t)(p.Hp / 2 + 3) * (p.Wp / 2 + 3) * 9 * C;
    p.n_Vp = (size_t)p.L * 25 * C;
    p.n_Ov = (size_t)(p.Hp + p.Wp) * 4 * C + C;
    p.Lld8 = (int)round_up(p.L, 8);
    p.n_P16 = HW * p.Lld8 / 2 + 64;
    return p;
}

}  // namespace ciaosr

using namespace ciaosr;

extern "C" size_t ciaosr_cs_attn_workspace_bytes(int H, int W, int C) { return ciaosr_cs_attn_workspace_bytes_scale(H, W, C, 2); }

extern "C" size_t ciaosr_cs_attn_workspace_bytes_scale(int H, int W, int C, int scale) {
    if (scale < 2 || scale > 4) scale = 4;          // callers sizing for "any scale" get the largest
    const CsaPlan p = csa_plan(H, W, C, scale);
    const size_t n = p.n_xp + p.n_E + p.n_M + p.n_x2 + p.n_R + p.n_Qp + p.n_Kn + p.n_V + p.n_S + p.n_O + p.n_Y + p.n_Yp +
                     2 * p.n_PE + p.n_Vp + p.n_Ov + p.n_P16;
    return n * sizeof(float) + 24 * 256;
}

static int cs_attn(const float* feat_hwc, int ld_feat, int H, int W, const ciaosr_csattn_weights_t* w, float* out, int ld_out,
                   const ciaosr_options_t* opt, void* workspace, size_t workspace_bytes, void* stream_, Prec prec) {
    CIAOSR_CHECK_ARG(feat_hwc && w && out && workspace && H >= 2 && W >= 2);
    CIAOSR_CHECK_ARG(options_ok(opt));
    const int C = w->channels;
    const int sc = w->scale ? w->scale : 2;
    CIAOSR_CHECK_ARG(C >= 4 && (C & 3) == 0 && ld_feat >= C && (ld_feat & 3) == 0 && (ld_out & 3) == 0);
    CIAOSR_CHECK_ARG(sc >= 2 && sc <= 4 && H >= sc && W >= sc);     // reflect padding needs pad < size
    hipStream_t s = (hipStream_t)stream_;
    const CsaPlan p = csa_plan(H, W, C, sc);
    if (workspace_bytes < ciaosr_cs_attn_workspace_bytes_scale(H, W, C, sc)) return CIAOSR_ERR_WORKSPACE;
    Arena ar(workspace, workspace_bytes);
    float* xp = ar.take<float>(p.n_xp);
    float* E = ar.take<float>(p.n_E);
    float* M = ar.take<float>(p.n_M);
    float* x2 = ar.take<float>(p.n_x2);
    float* R = ar.take<float>(p.n_R);
    float* Qp = ar.take<float>(p.n_Qp);
    float* Kn = ar.take<float>(p.n_Kn);
    float* V = ar.take<float>(p.n_V);
    float* S = ar.take<float>(p.n_S);
    float* O = ar.take<float>(p.n_O);
    float* Y = ar.take<float>(p.n_Y);
    float* Yp = ar.take<float>(p.n_Yp);
    float* PE = ar.take<float>(p.n_PE);
    float* Pc = ar.take<float>(p.n_PE);
    float* Vp = ar.take<float>(p.n_Vp);
    float* Ov = ar.take<float>(p.n_Ov);
    unsigned short* P16 = reinterpret_cast<unsigned short*>(ar.take<float>(p.n_P16));
    if (!ar.ok) return CIAOSR_ERR_WORKSPACE;

    const int HWp = p.Hp * p.Wp;
    int rc;
#define RUN(x) do { rc = (x); if (rc != CIAOSR_OK) return rc; } while (0)
    RUN(pad_reflect(feat_hwc, ld_feat, H, W, C, xp, p.Hp, p.Wp, s));
    // 1x1 convolutions + PReLU: the no-staging small GEMM on small maps
    auto conv1x1 = [&](const float* src, const float* wgt, const float* bias, float slope, float* dst, int n_out, int rows) -> int {
        if (gemm_small_ok(rows, n_out, C, C, C) && rows <= 4096)
            return gemm_small_f32(src, C, wgt, C, bias, dst, n_out, nullptr, 0, nullptr, 0, rows, n_out, C, CIAOSR_ACT_PRELU, slope, 1.f, s,
                                  "csa_conv1x1");
        return gemm_f32(src, C, wgt, C, false, dst, n_out, bias, rows, n_out, C, 1.f, CIAOSR_ACT_PRELU, slope, s, "csa_conv1x1");
    };
    RUN(conv1x1(xp, w->w_assembly, w->b_assembly, w->slope_assembly, E, C, HWp));
    RUN(conv1x1(xp, w->w_match1, w->b_match1, w->slope_match1, M, p.Ch, HWp));
    if (sc == 2) RUN(avgpool2(xp, p.Hp, p.Wp, C, x2, s));
    else RUN(downsample(xp, p.Hp, p.Wp, C, sc, x2, s));
    RUN(conv1x1(x2, w->w_match2, w->b_match2, w->slope_match2, R, p.Ch, p.L));
    // fp32: the correlation scores as a 3x3 diagonal box sum of the per-pixel correlation (csa_scores_f32.hip: K = 32 instead of 288,
    // no patch rows at all) -- per-call option csa_scores_gemm = 1 keeps the patch-row GEMM
    const bool box_scores = prec == kF32 && !(opt && opt->csa_scores_gemm) && csa_scores_box_ok(p.Ch, p.Ch, p.Ch) &&
                            (size_t)p.L <= p.n_Kn;
    if (!box_scores) {
        RUN(patch_rows(M, p.Ch, p.Hp, p.Wp, p.Ch, 3, 1, 1, p.Hp, p.Wp, Qp, 9 * p.Ch, 0, 0.f, s, "csa_patch_q"));
        RUN(patch_rows(R, p.Ch, p.Hp / sc, p.Wp / sc, p.Ch, 3, 1, 1, p.Hp / sc, p.Wp / sc, Kn, 9 * p.Ch, 1, w->escape_nan, s,
                       "csa_patch_k"));
    }
    // composed fold+down tail from this many (padded) LR pixels on: per-call option, default 4096
    const int composed_min = opt && opt->csa_composed_min ? opt->csa_composed_min : 4096;
    const bool composed = sc == 2 && w->w_down_masked && composed_min > 0 && HWp >= composed_min;     // the composed tail is scale 2's
    // 16-bit modes (big maps, composed tail): Q.K^T and P.V' on the bf16 / f16 MFMA (gemm_h16.hip); logits and softmax in fp32,
    // probabilities rounded to 16 bits; 1x1 convolutions, the partial down-convolutions and the final gather stay fp32
    const size_t qk16_bytes = ((size_t)HWp + p.L) * 9 * p.Ch * 2 + 512;
    if (prec != kF32 && composed && (9 * p.Ch) % 8 == 0 && (p.Lld & 3) == 0 &&
        qk16_bytes <= p.n_Y * sizeof(float) && (size_t)25 * C * p.Lld8 * 2 <= p.n_V * sizeof(float)) {
        const int Hh = p.Hp / 2, Wh = p.Wp / 2, Kq = 9 * p.Ch;
        const H16Ops& h = h16_ops(prec);
        const bool f16 = prec == kF16;
        unsigned short* Qb = reinterpret_cast<unsigned short*>(Y);
        unsigned short* Kb = Qb + round_up((size_t)HWp * Kq, 128);
        unsigned short* VpT = reinterpret_cast<unsigned short*>(V);
        RUN(h.cast_rows(Qp, Kq, Qb, Kq, HWp, Kq, s));
        RUN(h.cast_rows(Kn, Kq, Kb, Kq, p.L, Kq, s));
        // probabilities straight from the contraction (two passes over the short-K GEMM, no fp32 logit matrix, no softmax kernel);
        // the logits buffer serves as the statistics scratch
        // (the fused form takes its pass-1 maximum on the raw accumulators, which needs a positive scale; a non-positive
        // softmax_scale -- no config has one -- takes the logits + softmax_rows route instead of being refused)
        if (h.softmax_gemm_scratch(HWp, p.L) <= p.n_S && w->softmax_scale > 0.f) {
            RUN(h.softmax_gemm_nt(Qb, Kq, Kb, Kq, P16, p.Lld8, HWp, p.L, Kq, w->softmax_scale, S, p.n_S, s,
                                  f16 ? "csa_scores_f16" : "csa_scores_bf16"));
        } else {
            RUN(h.gemm_nt(Qb, Kq, Kb, Kq, S, p.Lld, false, HWp, p.L, Kq, w->softmax_scale, s, f16 ? "csa_scores_f16" : "csa_scores_bf16"));
            RUN(h.softmax_rows(S, HWp, p.L, p.Lld, P16, p.Lld8, s));
        }
        RUN(patch_rows(E, C, p.Hp, p.Wp, C, 3, 2, 3, Hh + 3, Wh + 3, PE, 9 * C, 0, 0.f, s, "csa_patch_down"));
        RUN(gemm_f32(PE, 9 * C, w->w_down_masked, 9 * C, false, Pc, 9 * C, nullptr, (Hh + 3) * (Wh + 3), 9 * C, 9 * C, 1.f,
                     CIAOSR_ACT_NONE, 0.f, s, "csa_down_partial"));
        RUN(csa_gather_vprime_t_h16(Pc, Hh, Wh, C, VpT, p.Lld8, f16, s));
        // main 16C columns for every row on the 16-bit MFMA
        RUN(h.gemm_nt(P16, p.Lld8, VpT, p.Lld8, O, 16 * C, false, HWp, 16 * C, p.Lld8, 1.f, s, f16 ? "csa_attn_v_f16" : "csa_attn_v_bf16"));
        // the 9C edge-variant columns are read for row 0 / column 0 pixels only (Wp + Hp of the HWp rows): those rows' 16-bit
        // probabilities go back to fp32 (exact) and through the fp32 path's three skinny split-K contractions
        float* Otop = Ov;
        float* Oleft = Ov + (size_t)p.Wp * 4 * C;
        float* Otl = Oleft + (size_t)p.Hp * 4 * C;
        float* Se = S;                                        // the logits are consumed: [Wp + Hp][Lld] fp32 rows fit
        float* part = reinterpret_cast<float*>(Y);            // so are the 16-bit Q / K copies
        RUN(csa_gather_vprime(Pc, Hh, Wh, C, Vp, s));
        RUN(h.rows_to_f32(P16, p.Lld8, 0, 1, p.Wp, p.Lld, Se, p.Lld, s));                               // row 0: pixels 0 .. Wp-1
        RUN(h.rows_to_f32(P16, p.Lld8, 0, p.Wp, p.Hp, p.Lld, Se + (size_t)p.Wp * p.Lld, p.Lld, s));     // column 0: pixels i * Wp
        RUN(gemm_f32_splitk(Se, p.Lld, Vp + 16 * C, 25 * C, true, Otop, 4 * C, nullptr, p.Wp, 4 * C, p.L, 1.f, CIAOSR_ACT_NONE,
                            0.f, part, p.n_Y, s, "csa_attn_v_edge"));
        RUN(gemm_f32_splitk(Se + (size_t)p.Wp * p.Lld, p.Lld, Vp + 20 * C, 25 * C, true, Oleft, 4 * C, nullptr, p.Hp, 4 * C, p.L, 1.f,
                            CIAOSR_ACT_NONE, 0.f, part, p.n_Y, s, "csa_attn_v_edge"));
        RUN(gemm_f32_splitk(Se, p.Lld, Vp + 24 * C, 25 * C, true, Otl, C, nullptr, 1, C, p.L, 1.f, CIAOSR_ACT_NONE, 0.f, part,
                            p.n_Y, s, "csa_attn_v_edge"));
        RUN(csa_gather_out(O, Otop, Oleft, Otl, w->b_down, H, W, p.Hp, p.Wp, C, out, ld_out, 16L * C, 4L * C, 4L * C, s));
        return CIAOSR_OK;
    }
    if (box_scores)
        RUN(csa_scores_box_f32(M, p.Ch, p.Hp, p.Wp, R, p.Ch, p.Hp / sc, p.Wp / sc, p.Ch, w->softmax_scale, w->escape_nan, Kn /*norms*/, S,
                               p.Lld, s));
    else
        RUN(gemm_f32(Qp, 9 * p.Ch, Kn, 9 * p.Ch, false, S, p.Lld, nullptr, HWp, p.L, 9 * p.Ch, w->softmax_scale,
                     CIAOSR_ACT_NONE, 0.f, s, "csa_scores"));
    if (composed && (size_t)HWp * 2 <= p.n_Qp) {
        // composed fold + down with the row softmax applied in the attn.V operand staging (statistics-only pass over S: the in-place
        // rewrite of the 1.36-GB logit matrix is gone; probabilities = exp2(x log2 e - max log2 e) / sum, equal to softmax_rows' to rounding)
        const int Hh = p.Hp / 2, Wh = p.Wp / 2;
        float* st = Qp;                                       // [HWp] (max x log2 e, 1 / sum); the patch rows are consumed (or were never built)
        float* Otop = Ov;
        float* Oleft = Ov + (size_t)p.Wp * 4 * C;
        float* Otl = Oleft + (size_t)p.Hp * 4 * C;
        RUN(softmax_stats_rows(S, HWp, p.L, p.Lld, st, s));
        RUN(patch_rows(E, C, p.Hp, p.Wp, C, 3, 2, 3, Hh + 3, Wh + 3, PE, 9 * C, 0, 0.f, s, "csa_patch_down"));
        RUN(gemm_f32(PE, 9 * C, w->w_down_masked, 9 * C, false, Pc, 9 * C, nullptr, (Hh + 3) * (Wh + 3), 9 * C, 9 * C, 1.f,
                     CIAOSR_ACT_NONE, 0.f, s, "csa_down_partial"));
        RUN(csa_gather_vprime(Pc, Hh, Wh, C, Vp, s));
        // attn.V: at a C3 tile's size (768 tiles of 192 x 256) one workgroup per CU, else -- or on request -- the 128 x 128 kernel; bitwise equal
        if (!(opt && opt->csa_attn_tile128) && gemm_big_softmax_f32_ok(p.Lld, 25 * C, HWp, 16 * C, p.L, true) &&
            ((size_t)(HWp - 1) * p.Lld + p.L) * sizeof(float) < 0xFFFFFF00ull)
            RUN(gemm_big_softmax_f32(S, p.Lld, st, 1, Vp, 25 * C, O, 16 * C, HWp, 16 * C, p.L, s, "csa_attn_v"));
        else
            RUN(gemm_f32_softmax_a(S, p.Lld, st, 1, Vp, 25 * C, true, O, 16 * C, HWp, 16 * C, p.L, nullptr, 0, s, "csa_attn_v"));
        RUN(gemm_f32_softmax_a(S, p.Lld, st, 1, Vp + 16 * C, 25 * C, true, Otop, 4 * C, p.Wp, 4 * C, p.L, Y, p.n_Y, s, "csa_attn_v_edge"));
        RUN(gemm_f32_softmax_a(S, p.Wp * p.Lld, st, p.Wp, Vp + 20 * C, 25 * C, true, Oleft, 4 * C, p.Hp, 4 * C, p.L, Y, p.n_Y, s,
                               "csa_attn_v_edge"));
        RUN(gemm_f32_softmax_a(S, p.Lld, st, 1, Vp + 24 * C, 25 * C, true, Otl, C, 1, C, p.L, Y, p.n_Y, s, "csa_attn_v_edge"));
        RUN(csa_gather_out(O, Otop, Oleft, Otl, w->b_down, H, W, p.Hp, p.Wp, C, out, ld_out, 16L * C, 4L * C, 4L * C, s));
        return CIAOSR_OK;
    }
    RUN(softmax_rows(S, HWp, p.L, p.Lld, s));
    if (composed) {
        // composed fold + down (patch_ops.hip): attn.V with N = 16C instead of 36C, no 2x map, no separate down conv
        const int Hh = p.Hp / 2, Wh = p.Wp / 2;
        float* Otop = Ov;
        float* Oleft = Ov + (size_t)p.Wp * 4 * C;
        float* Otl = Oleft + (size_t)p.Hp * 4 * C;
        RUN(patch_rows(E, C, p.Hp, p.Wp, C, 3, 2, 3, Hh + 3, Wh + 3, PE, 9 * C, 0, 0.f, s, "csa_patch_down"));
        RUN(gemm_f32(PE, 9 * C, w->w_down_masked, 9 * C, false, Pc, 9 * C, nullptr, (Hh + 3) * (Wh + 3), 9 * C, 9 * C, 1.f,
                     CIAOSR_ACT_NONE, 0.f, s, "csa_down_partial"));
        RUN(csa_gather_vprime(Pc, Hh, Wh, C, Vp, s));
        RUN(gemm_f32(S, p.Lld, Vp, 25 * C, true, O, 16 * C, nullptr, HWp, 16 * C, p.L, 1.f, CIAOSR_ACT_NONE, 0.f, s,
                     "csa_attn_v"));
        // the three edge variants are skinny (M = Wp, Hp, 1 rows; K = L): split-K over the (now free) 2x-map buffer
        RUN(gemm_f32_splitk(S, p.Lld, Vp + 16 * C, 25 * C, true, Otop, 4 * C, nullptr, p.Wp, 4 * C, p.L, 1.f, CIAOSR_ACT_NONE,
                            0.f, Y, p.n_Y, s, "csa_attn_v_edge"));
        RUN(gemm_f32_splitk(S, p.Wp * p.Lld, Vp + 20 * C, 25 * C, true, Oleft, 4 * C, nullptr, p.Hp, 4 * C, p.L, 1.f,
                            CIAOSR_ACT_NONE, 0.f, Y, p.n_Y, s, "csa_attn_v_edge"));
        RUN(gemm_f32_splitk(S, p.Lld, Vp + 24 * C, 25 * C, true, Otl, C, nullptr, 1, C, p.L, 1.f, CIAOSR_ACT_NONE, 0.f, Y,
                            p.n_Y, s, "csa_attn_v_edge"));
        RUN(csa_gather_out(O, Otop, Oleft, Otl, w->b_down, H, W, p.Hp, p.Wp, C, out, ld_out, 16L * C, 4L * C, 4L * C, s));
        return CIAOSR_OK;
    }
    // V patches (3s)x(3s), stride s, 'same' padding = s each side (csa:462-465); attn.V; conv_transpose2d(stride s, padding s) as a
    // gather; the scale's down conv (3x3, stride s, pad 1: down / downx3 / downx4, csa:516-521) on the cropped H x W outputs
    const int kv = 9 * sc * sc * C;
    RUN(patch_rows(E, C, p.Hp, p.Wp, C, 3 * sc, sc, sc, p.Hp / sc, p.Wp / sc, V, kv, 0, 0.f, s, "csa_patch_v"));
    RUN(gemm_f32(S, p.Lld, V, kv, true, O, kv, nullptr, HWp, kv, p.L, 1.f, CIAOSR_ACT_NONE, 0.f, s, "csa_attn_v"));
    if (sc == 2) RUN(fold(O, kv, p.Hp, p.Wp, C, Y, s));
    else RUN(fold_s(O, kv, p.Hp, p.Wp, C, sc, Y, s));
    RUN(patch_rows(Y, C, sc * p.Hp, sc * p.Wp, C, 3, sc, 1, H, W, Yp, 9 * C, 0, 0.f, s, "csa_patch_down"));
    if (gemm_small_ok(H * W, C, 9 * C, 9 * C, 9 * C) && H * W <= 4096)
        RUN(gemm_small_f32(Yp, 9 * C, w->w_down, 9 * C, w->b_down, out, ld_out, nullptr, 0, nullptr, 0, H * W, C, 9 * C, CIAOSR_ACT_NONE,
                           0.f, 1.0f / 6.0f, s, "csa_down"));
    else
        RUN(gemm_f32(Yp, 9 * C, w->w_down, 9 * C, false, out, ld_out, w->b_down, H * W, C, 9 * C, 1.0f / 6.0f,
                     CIAOSR_ACT_NONE, 0.f, s, "csa_down"));
#undef RUN
    return CIAOSR_OK;
}

extern "C" int ciaosr_cs_attn_f32(const float* feat_hwc, int ld_feat, int H, int W, const ciaosr_csattn_weights_t* w,
                                  float* out, int ld_out, const ciaosr_options_t* opt, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    return cs_attn(feat_hwc, ld_feat, H, W, w, out, ld_out, opt, workspace, workspace_bytes, stream, kF32);
}

extern "C" int ciaosr_cs_attn_bf16(const float* feat_hwc, int ld_feat, int H, int W, const ciaosr_csattn_weights_t* w,
                                   float* out, int ld_out, const ciaosr_options_t* opt, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    return cs_attn(feat_hwc, ld_feat, H, W, w, out, ld_out, opt, workspace, workspace_bytes, stream, kBF16);
}

extern "C" int ciaosr_cs_attn_f16(const float* feat_hwc, int ld_feat, int H, int W, const ciaosr_csattn_weights_t* w,
                                  float* out, int ld_out, const ciaosr_options_t* opt, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    return cs_attn(feat_hwc, ld_feat, H, W, w, out, ld_out, opt, workspace, workspace_bytes, stream, kF16);
}
