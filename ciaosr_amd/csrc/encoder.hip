// Encoder trunks (`gen_feature`) as sequences of implicit-GEMM convolutions on channels-last maps.
//   RDN  : ciaosr_net.py:321-342 over mmedit's RDN modules (sfe1, sfe2, rdbs[b].layers[l].conv, rdbs[b].lff, gff)
//   EDSR : ciaosr_net.py:393-408 (conv_first, body[b].conv1/conv2, conv_after_body)
// Output: feature map [H][W][C] channels-last, exactly what the head consumes.
#include "ops.h"

namespace ciaosr {

int conv2d_hwc(const float* src, int ld_src, int H, int W, int Cin, const float* wgt, int ldw, const float* bias,
               int Cout, int ksize, float* dst, int ld_dst, float* dst2, int ld_dst2, const float* res, int ld_res,
               int act, float alpha, float* partial, size_t partial_floats, hipStream_t s, const char* tag);

int dense_scatter_step(float* X, int ldx, int H, int W, int step, int num_layers, const float* wgt, const float* bias_all,
                       float* acc_buf, float* partial, size_t partial_floats, hipStream_t s);

// dense_scatter_f32.hip
int dense_scatter_small(float* X, int ldx, int H, int W, int step, int num_layers, const float* frag, const float* bias_all,
                        float* acc_buf, hipStream_t s);
// dense_f32.hip
int dense_f32_tiles(int H, int W);
int dense_layer_f32(float* X, int ldx, int H, int W, int l, const float* frag, const float* bias, int n_img, hipStream_t s);

__global__ void image_to_hwc4_kernel(const float* __restrict__ x, float* __restrict__ out, long HW) {
    // [3][H][W] -> [H*W][4] with a zero 4th channel (so the first conv moves float4 taps)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<float4*>(out)[i] = make_float4(x[i], x[HW + i], x[2 * HW + i], 0.f);
}

// first conv: 3 (padded to 4) input channels -> explicit patch rows [HW][36] + MFMA GEMM
static int first_conv(const float* x_nchw, int H, int W, const ciaosr_conv_t& c, float* img4, float* rows,
                      float* dst, int ld_dst, hipStream_t s) {
    const long HW = (long)H * W;
    {
        ProfScope prof("image_to_hwc4", s);
        int grid = (int)((HW + 255) / 256);
        hipLaunchKernelGGL(image_to_hwc4_kernel, dim3(grid > 2048 ? 2048 : grid), dim3(256), 0, s, x_nchw, img4, HW);
    }
    int rc = launch_status("image_to_hwc4");
    if (rc != CIAOSR_OK) return rc;
    rc = patch_rows(img4, 4, H, W, 4, 3, 1, 1, H, W, rows, 36, 0, 0.f, s, "enc_patch_first");
    if (rc != CIAOSR_OK) return rc;
    return gemm_f32(rows, 36, c.weight, 36, false, dst, ld_dst, c.bias, (int)HW, c.cout, 36, 1.f, CIAOSR_ACT_NONE, 0.f,
                    s, "enc_conv_first");
}

static bool conv_ok(const ciaosr_conv_t& c, int cin, int cout, int k) {
    return c.weight && c.bias && c.cin == cin && c.cout == cout && c.ksize == k;
}

// 3x3 trunk convolution: the one-launch small-map kernel when the layer has fragment-packed weights, else the generic path
static int conv3(const float* src, int ld_src, int H, int W, const ciaosr_conv_t& c, float* dst, int ld_dst, const float* res,
                 int ld_res, int act, float alpha, float* part, size_t pf, hipStream_t s) {
    if (c.frag && conv3x3_small_ok(H, W, c.cin, c.cout, ld_src, act))
        return conv3x3_small(src, ld_src, H, W, c.cin, c.frag, c.bias, c.cout, dst, ld_dst, nullptr, 0, res, ld_res, act, alpha, s,
                             "enc_conv3x3");
    return conv2d_hwc(src, ld_src, H, W, c.cin, c.weight, 9 * c.cin, c.bias, c.cout, 3, dst, ld_dst, nullptr, 0, res, ld_res, act,
                      alpha, part, pf, s, "enc_conv3x3");
}

}  // namespace ciaosr

using namespace ciaosr;

extern "C" size_t ciaosr_rdn_workspace_bytes_batch(int B, int H, int W, const ciaosr_rdn_weights_t* w) {
    if (!w || B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t HW = (size_t)H * W, BHW = (size_t)B * HW;
    const int C = w->mid_channels, G = w->growth, cb = C + G * w->num_layers;
    size_t n = HW * 4 + HW * 36 /*first-conv temporaries, one image at a time*/ + BHW * C /*sfe1*/ + 2 * BHW * cb /*block buffers*/ +
               BHW * (size_t)G * w->num_blocks /*global concat*/ + BHW * C /*gff0*/ + HW * (size_t)G * w->num_layers /*scatter sums*/ +
               16 * HW * (size_t)(C > G ? C : G) + BHW * cb / 2 + 64 /*16-bit copy of one block buffer*/ +
               (size_t)w->num_blocks * G * cb / 2 + 64 /*16-bit copies of the lff weights (f16 mode)*/;
    return n * sizeof(float) + 17 * 256;
}

extern "C" size_t ciaosr_rdn_workspace_bytes(int H, int W, const ciaosr_rdn_weights_t* w) {
    return ciaosr_rdn_workspace_bytes_batch(1, H, W, w);
}

// B images of the same size through the trunk.  On the big-map routes (halo-resident dense layers) the B images share every dense-layer
// launch (grid.y = image: the 128 strictly dependent launches per image pay their ~8.5 us ramp / first-load / K-slice-reduction / drain
// once per batch instead of once per image) and the row-wise 1x1 kernels of the f16 route; the few 3x3 convolutions outside the
// blocks run per image.  Each image is computed by exactly the workgroups, in exactly the order, of a single-image call: bitwise equal.
static int rdn_forward(const float* x_nchw, int B, int H, int W, const ciaosr_rdn_weights_t* w, float* feat_hwc,
                       const ciaosr_options_t* opt, void* workspace, size_t workspace_bytes, void* stream_, Prec prec) {
    // f16_pairs = 2 ("f16x3", the fp32-tolerance fast mode): the trunk runs its fp32 route -- half ACTIVATIONS in 128 dense layers
    // alone cost rms 4.6e-5 / max 4e-4 on the full C3 tile, and activation pairs (three MFMAs per product + a second patch) would
    // cost the dense layers about what the fp32 Winograd form does
    if ((prec == kF16 || prec == kBF16) && opt && opt->f16_pairs == 2) prec = kF32;      // "f16x3" and (round 6) "bf16x3"
    const bool bf16 = prec != kF32;      // a 16-bit MFMA mode (bf16 or f16 entry)
    // route thresholds (per-call options; defaults: halo-resident dense layers from 128 tiles of 12x12 pixels on, small-map
    // kernels up to 18432 pixels = 128 such tiles)
    const int min_tiles = opt && opt->dense_min_tiles ? opt->dense_min_tiles : 128;
    const int small_max = opt && opt->scatter_small_max ? opt->scatter_small_max : 18432;
    CIAOSR_CHECK_ARG(x_nchw && w && feat_hwc && workspace && B >= 1 && H > 0 && W > 0);
    CIAOSR_CHECK_ARG(options_ok(opt));
    const int C = w->mid_channels, G = w->growth, NB = w->num_blocks, NL = w->num_layers;
    CIAOSR_CHECK_ARG(C % 32 == 0 && G % 32 == 0 && NB >= 1 && NL >= 1 && w->dense && w->lff);
    CIAOSR_CHECK_ARG(C == G);   // mmedit's RDN feeds rdbs[b>0] with channel_growth channels and adds sfe1 (mid) at the end
    CIAOSR_CHECK_ARG(conv_ok(w->sfe1, 3, C, 3) && conv_ok(w->sfe2, C, C, 3));
    CIAOSR_CHECK_ARG(conv_ok(w->gff0, G * NB, C, 1) && conv_ok(w->gff1, C, C, 3));
    if (workspace_bytes < ciaosr_rdn_workspace_bytes_batch(B, H, W, w)) return CIAOSR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream_;
    const size_t HW = (size_t)H * W, BHW = (size_t)B * HW;
    const int cb = C + G * NL;
    // The batched block buffers [B*HW][cb] are addressed with 32-bit buffer offsets by the halo-resident dense kernels: a batch that
    // does not fit runs as sub-batches that do (same workgroups per image: still bitwise the one-image result); a SINGLE image that
    // does not fit leaves the halo-resident routes to the generic ones, whose launchers check their own operands.
    const size_t widest = (size_t)(cb > G * NB ? cb : G * NB);      // block buffer or global concat rows, whichever is wider
    const bool fits32 = BHW * widest * sizeof(float) < 0xFFFFFF00ull;
    if (B > 1 && !fits32) {
        size_t bmax = (size_t)(0xFFFFFF00ull - 1) / (HW * widest * sizeof(float));
        if (bmax < 1) bmax = 1;
        for (int i = 0; i < B; i += (int)bmax) {
            const int nb = B - i < (int)bmax ? B - i : (int)bmax;
            const int rc = rdn_forward(x_nchw + (size_t)i * 3 * HW, nb, H, W, w, feat_hwc + (size_t)i * HW * C, opt, workspace, workspace_bytes,
                                       stream_, prec);
            if (rc != CIAOSR_OK) return rc;
        }
        return CIAOSR_OK;
    }
    Arena ar(workspace, workspace_bytes);
    float* img4 = ar.take<float>(HW * 4);
    float* rows = ar.take<float>(HW * 36);
    float* sfe1 = ar.take<float>(BHW * C);
    float* X[2] = {ar.take<float>(BHW * cb), ar.take<float>(BHW * cb)};
    float* Gc = ar.take<float>(BHW * (size_t)G * NB);
    float* g0 = ar.take<float>(BHW * C);
    float* accb = ar.take<float>(HW * (size_t)G * NL);
    const size_t pf = 16 * HW * (size_t)(C > G ? C : G);
    float* part = ar.take<float>(pf);
    unsigned short* Xb = reinterpret_cast<unsigned short*>(ar.take<float>(BHW * cb / 2 + 64));
    unsigned short* Wl16 = reinterpret_cast<unsigned short*>(ar.take<float>((size_t)NB * G * cb / 2 + 64));
    if (!ar.ok) return CIAOSR_ERR_WORKSPACE;
    // 16-bit modes: the dense layers (97 % of the trunk's MACs) run on the bf16 / f16 MFMA when the map is big enough to give
    // every CU a tile (dense_h16.hip); first/last convolutions, LFF/GFF 1x1 and all residual sums stay fp32
    bool dense16 = fits32 && bf16 && C == 64 && G == 64 && min_tiles > 0 && b16::dense_h16_tiles(H, W) >= min_tiles;
    if (bf16)
        for (int i = 0; i < NB * NL && dense16; ++i) dense16 = w->dense[i].frag16 != nullptr;
    // big maps, fp32: halo-resident gather-form dense layers (dense_f32.hip) instead of the scatter form
    bool dense32 = fits32 && !dense16 && C == 64 && G == 64 && min_tiles > 0 && dense_f32_tiles(H, W) >= min_tiles;
    for (int i = 0; i < NB * NL && dense32; ++i) dense32 = w->dense[i].frag != nullptr;
    // ... in Winograd F(2x2, 3x3) form when the transformed weights are there (2.25x fewer MFMAs; dense_wino_f32.hip)
    const int dd = opt ? opt->dense_direct : 0;       // 0 = best Winograd form available, 1 = direct, 2 = F(2x2)
    bool wino32 = dense32 && dd != 1;
    for (int i = 0; i < NB * NL && wino32; ++i) wino32 = w->dense[i].frag_wino != nullptr;
    // ... or F(4x4, 3x3): 4x fewer MFMAs than the direct form (dense_wino4_f32.hip)
    bool wino4 = dense32 && dd == 0;
    for (int i = 0; i < NB * NL && wino4; ++i) wino4 = w->dense[i].frag_wino4 != nullptr;
    // f16 mode: the local feature fusion (1x1 over the block's 576 channels) too reads the 16-bit copy of the block buffer, on the
    // 16-bit GEMM with bias + residual in its epilogue; the dense layers then need no fp32 copy of their outputs, and the epilogue
    // writes the next block's 16-bit input group.  (bf16 mode keeps the fp32 lff: its weights would need the hi + lo pair.)
    const bool pairs16 = prec == kF16 && opt && opt->f16_pairs;       // half weight pairs: the lff keeps its fp32 weights, like the bf16 mode
    const bool lff16 = dense16 && prec == kF16 && !pairs16 && cb % 8 == 0 && G % 4 == 0 && G <= 128;
    int rc;
#define RUN(x) do { rc = (x); if (rc != CIAOSR_OK) return rc; } while (0)
    if (B > 1 && !(dense16 || dense32)) {       // small maps: one image after the other through the single-image routes
        for (int i = 0; i < B; ++i)
            RUN(rdn_forward(x_nchw + (size_t)i * 3 * HW, 1, H, W, w, feat_hwc + (size_t)i * HW * C, opt, workspace, workspace_bytes, stream_, prec));
        return CIAOSR_OK;
    }
    if (lff16) {
        const float* src[16];
        for (int b0 = 0; b0 < NB; b0 += 16) {
            const int n = NB - b0 < 16 ? NB - b0 : 16;
            for (int i = 0; i < n; ++i) {
                CIAOSR_CHECK_ARG(conv_ok(w->lff[b0 + i], cb, G, 1));
                src[i] = w->lff[b0 + i].weight;
            }
            RUN(h16_ops(prec).cast_many(src, n, G, cb, Wl16 + (size_t)b0 * G * cb, s));
        }
    }
    for (int i = 0; i < B; ++i) {
        RUN(first_conv(x_nchw + (size_t)i * 3 * HW, H, W, w->sfe1, img4, rows, sfe1 + (size_t)i * HW * C, C, s));
        // sfe2 -> block 0 input (columns [0, C) of X[0])
        RUN(conv3(sfe1 + (size_t)i * HW * C, C, H, W, w->sfe2, X[0] + (size_t)i * HW * cb, cb, nullptr, 0, CIAOSR_ACT_NONE, 1.f, part, pf, s));
    }
    for (int b = 0; b < NB; ++b) {
        float* x = X[b & 1];
        float* xn = X[(b + 1) & 1];
        if (dense16) {
            if (!(lff16 && b > 0)) RUN(h16_ops(prec).cast_group(x, cb, Xb, cb, 0, (long)BHW, s));     // else: written by the previous lff
            for (int l = 0; l < NL; ++l) {
                const ciaosr_conv_t& c = w->dense[b * NL + l];
                CIAOSR_CHECK_ARG(conv_ok(c, C + G * l, G, 3));
                RUN(h16_ops(prec).dense_layer(lff16 ? nullptr : x, cb, Xb, cb, H, W, l, c.frag16,
                                              (prec == kF16 ? !pairs16 : (opt && opt->bf16_single)) ? nullptr : c.frag16_lo, c.bias, B, s,
                                              dd == 1 ? 1 : 0));
            }
            if (lff16) {
                // RDB output = x + lff(dense) from the 16-bit rows: fp32 to the global concat and the next block's input, 16-bit to the
                // next block's input group (rows of Xb this workgroup alone reads and writes: N = G is one column tile)
                const bool more = b + 1 < NB;
                RUN(h16_ops(prec).conv1x1(Xb, cb, Wl16 + (size_t)b * G * cb, cb, w->lff[b].bias, x, cb, Gc + (size_t)b * G, G * NB,
                                          more ? xn : nullptr, cb, more ? Xb : nullptr, cb, (int)BHW, G, cb, s, "enc_conv1x1_f16"));
                continue;
            }
        } else if (dense32) {
            for (int l = 0; l < NL; ++l) {
                const ciaosr_conv_t& c = w->dense[b * NL + l];
                CIAOSR_CHECK_ARG(conv_ok(c, C + G * l, G, 3));
                if (wino4) RUN(dense_layer_wino4_f32(x, cb, H, W, l, c.frag_wino4, c.bias, B, s));
                else if (wino32) RUN(dense_layer_wino_f32(x, cb, H, W, l, c.frag_wino, c.bias, B, s));
                else RUN(dense_layer_f32(x, cb, H, W, l, c.frag, c.bias, B, s));
            }
        } else if (w->scatter_weight && w->scatter_bias && C == 64 && G == 64) {
            // scatter form: input group s (64 channels) feeds every later dense layer in ONE convolution with
            // N = 64*(NL-s) output channels and K = 576: no split-K slabs, 8 launches instead of 16
            const bool small = w->scatter_frag && (long)HW <= small_max;
            for (int st = 0; st < NL; ++st) {
                if (small && w->scatter_frag[b * NL + st])
                    RUN(dense_scatter_small(x, cb, H, W, st, NL, w->scatter_frag[b * NL + st], w->scatter_bias + (size_t)b * NL * 64,
                                            accb, s));
                else
                    RUN(dense_scatter_step(x, cb, H, W, st, NL, w->scatter_weight[b * NL + st],
                                           w->scatter_bias + (size_t)b * NL * 64, accb, part, pf, s));
            }
        } else {
            for (int l = 0; l < NL; ++l) {
                const ciaosr_conv_t& c = w->dense[b * NL + l];
                const int cin = C + G * l;
                CIAOSR_CHECK_ARG(conv_ok(c, cin, G, 3));
                // DenseLayer: cat([x, relu(conv(x))]) == write the G new channels next to the inputs
                RUN(conv2d_hwc(x, cb, H, W, cin, c.weight, 9 * cin, c.bias, G, 3, x + cin, cb, nullptr, 0, nullptr, 0,
                               CIAOSR_ACT_RELU, 1.f, part, pf, s, "enc_conv3x3"));
            }
        }
        const ciaosr_conv_t& f = w->lff[b];
        CIAOSR_CHECK_ARG(conv_ok(f, cb, G, 1));
        // RDB output = x + lff(dense): goes to the global concat and is the next block's input
        if (conv1x1_resident_ok((long)HW, G, cb, cb, cb) && fits32) {
            // big maps: the whole batch in ONE launch of the weights-resident kernel (the B images' rows are contiguous in every buffer)
            RUN(conv1x1_resident_f32(x, cb, f.weight, cb, f.bias, x, cb, Gc + (size_t)b * G, G * NB, b + 1 < NB ? xn : nullptr, cb, (long)BHW, cb,
                                     s, "enc_conv1x1"));
            continue;
        }
        for (int i = 0; i < B; ++i) {
            float* xi = x + (size_t)i * HW * cb;
            float* xni = b + 1 < NB ? xn + (size_t)i * HW * cb : nullptr;
            float* gi = Gc + (size_t)i * HW * G * NB + (size_t)b * G;
            if (gemm_small_ok((int)HW, G, cb, cb, cb))
                RUN(gemm_small_f32(xi, cb, f.weight, cb, f.bias, gi, G * NB, xni, cb, xi, cb, (int)HW, G, cb, CIAOSR_ACT_NONE, 0.f, 1.f, s,
                                   "enc_conv1x1"));
            else
                RUN(conv2d_hwc(xi, cb, H, W, cb, f.weight, cb, f.bias, G, 1, gi, G * NB, xni, cb, xi, cb, CIAOSR_ACT_NONE, 1.f, part, pf, s,
                               "enc_conv1x1"));
        }
    }
    // global feature fusion; its own profiler tag when the blocks' 1x1 convolutions ran on the 16-bit path (the "enc_conv1x1" work
    // figure of bench.py counts both)
    const char* gff_tag = lff16 ? "enc_gff1x1" : "enc_conv1x1";
    for (int i = 0; i < B; ++i) {
        const float* gci = Gc + (size_t)i * HW * G * NB;
        float* g0i = g0 + (size_t)i * HW * C;
        if (gemm_small_ok((int)HW, C, G * NB, G * NB, G * NB))
            RUN(gemm_small_f32(gci, G * NB, w->gff0.weight, G * NB, w->gff0.bias, g0i, C, nullptr, 0, nullptr, 0, (int)HW, C, G * NB,
                               CIAOSR_ACT_NONE, 0.f, 1.f, s, gff_tag));
        else
            RUN(conv2d_hwc(gci, G * NB, H, W, G * NB, w->gff0.weight, G * NB, w->gff0.bias, C, 1, g0i, C, nullptr, 0, nullptr, 0,
                           CIAOSR_ACT_NONE, 1.f, part, pf, s, gff_tag));
        RUN(conv3(g0i, C, H, W, w->gff1, feat_hwc + (size_t)i * HW * C, C, sfe1 + (size_t)i * HW * C, C, CIAOSR_ACT_NONE, 1.f, part, pf, s));
    }
#undef RUN
    return CIAOSR_OK;
}

extern "C" int ciaosr_rdn_forward_f32(const float* x_nchw, int H, int W, const ciaosr_rdn_weights_t* w,
                                      float* feat_hwc, const ciaosr_options_t* opt, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    return rdn_forward(x_nchw, 1, H, W, w, feat_hwc, opt, workspace, workspace_bytes, stream, kF32);
}

extern "C" int ciaosr_rdn_forward_batch_f32(const float* x_nchw, int B, int H, int W, const ciaosr_rdn_weights_t* w,
                                              float* feat_hwc, const ciaosr_options_t* opt, void* workspace,
                                              size_t workspace_bytes, void* stream) {
    return rdn_forward(x_nchw, B, H, W, w, feat_hwc, opt, workspace, workspace_bytes, stream, kF32);
}

extern "C" int ciaosr_rdn_forward_bf16(const float* x_nchw, int H, int W, const ciaosr_rdn_weights_t* w,
                                       float* feat_hwc, const ciaosr_options_t* opt, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return rdn_forward(x_nchw, 1, H, W, w, feat_hwc, opt, workspace, workspace_bytes, stream, kBF16);
}

extern "C" int ciaosr_rdn_forward_batch_bf16(const float* x_nchw, int B, int H, int W, const ciaosr_rdn_weights_t* w,
                                              float* feat_hwc, const ciaosr_options_t* opt, void* workspace,
                                              size_t workspace_bytes, void* stream) {
    return rdn_forward(x_nchw, B, H, W, w, feat_hwc, opt, workspace, workspace_bytes, stream, kBF16);
}

extern "C" int ciaosr_rdn_forward_f16(const float* x_nchw, int H, int W, const ciaosr_rdn_weights_t* w,
                                      float* feat_hwc, const ciaosr_options_t* opt, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    return rdn_forward(x_nchw, 1, H, W, w, feat_hwc, opt, workspace, workspace_bytes, stream, kF16);
}

extern "C" int ciaosr_rdn_forward_batch_f16(const float* x_nchw, int B, int H, int W, const ciaosr_rdn_weights_t* w,
                                              float* feat_hwc, const ciaosr_options_t* opt, void* workspace,
                                              size_t workspace_bytes, void* stream) {
    return rdn_forward(x_nchw, B, H, W, w, feat_hwc, opt, workspace, workspace_bytes, stream, kF16);
}

extern "C" size_t ciaosr_edsr_workspace_bytes(int H, int W, const ciaosr_edsr_weights_t* w) {
    if (!w || H <= 0 || W <= 0) return 0;
    const size_t HW = (size_t)H * W;
    const int C = w->mid_channels;
    return (HW * 4 + HW * 36 + 4 * HW * C + 16 * HW * C) * sizeof(float) + 16 * 256;
}

extern "C" int ciaosr_edsr_forward_f32(const float* x_nchw, int H, int W, const ciaosr_edsr_weights_t* w,
                                       float* feat_hwc, void* workspace, size_t workspace_bytes, void* stream_) {
    CIAOSR_CHECK_ARG(x_nchw && w && feat_hwc && workspace && H > 0 && W > 0);
    const int C = w->mid_channels, NB = w->num_blocks;
    CIAOSR_CHECK_ARG(C % 32 == 0 && NB >= 0 && (NB == 0 || (w->conv1 && w->conv2)));
    CIAOSR_CHECK_ARG(conv_ok(w->conv_first, 3, C, 3) && conv_ok(w->conv_after_body, C, C, 3));
    if (workspace_bytes < ciaosr_edsr_workspace_bytes(H, W, w)) return CIAOSR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream_;
    const size_t HW = (size_t)H * W;
    Arena ar(workspace, workspace_bytes);
    float* img4 = ar.take<float>(HW * 4);
    float* rows = ar.take<float>(HW * 36);
    float* first = ar.take<float>(HW * C);
    float* a = ar.take<float>(HW * C);
    float* b = ar.take<float>(HW * C);
    float* tmp = ar.take<float>(HW * C);
    const size_t pf = 16 * HW * (size_t)C;
    float* part = ar.take<float>(pf);
    if (!ar.ok) return CIAOSR_ERR_WORKSPACE;
    int rc;
#define RUN(x) do { rc = (x); if (rc != CIAOSR_OK) return rc; } while (0)
    RUN(first_conv(x_nchw, H, W, w->conv_first, img4, rows, first, C, s));
    const float* cur = first;
    float* pp[2] = {a, b};
    for (int i = 0; i < NB; ++i) {
        CIAOSR_CHECK_ARG(conv_ok(w->conv1[i], C, C, 3) && conv_ok(w->conv2[i], C, C, 3));
        // ResidualBlockNoBN: x + conv2(relu(conv1(x))) * res_scale
        RUN(conv3(cur, C, H, W, w->conv1[i], tmp, C, nullptr, 0, CIAOSR_ACT_RELU, 1.f, part, pf, s));
        RUN(conv3(tmp, C, H, W, w->conv2[i], pp[i & 1], C, cur, C, CIAOSR_ACT_NONE, w->res_scale, part, pf, s));
        cur = pp[i & 1];
    }
    RUN(conv3(cur, C, H, W, w->conv_after_body, feat_hwc, C, first, C, CIAOSR_ACT_NONE, 1.f, part, pf, s));
#undef RUN
    return CIAOSR_OK;
}
