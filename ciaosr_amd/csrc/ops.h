// Internal declarations shared by the translation units of libciaosr_hip.so.
#pragma once
#include "common.h"

namespace ciaosr {

constexpr int MAXJ = 9;  // key samples per query for local_size 3

struct HeadRowsP {
    const float* coord;
    const float* cell;
    long q0;
    int nq, chunk, H, W, local_size, J;
    const float* Tk;   // [HW][wk0]  layer-0 table of imnet_k (bias included)
    const float* Tv;   // [HW][wv0]
    const float* tailK;  // &W1k[0][D]: 4 tail columns (rel_y rel_x scale_y scale_x) of each row, stride ld_tail_k
    const float* tailV;  // &W1v[0][Dv]
    int ld_tail_k, ld_tail_v;
    int wk0, wv0, relu_k, relu_v;   // relu_*: CIAOSR_ACT_* code of the layer-0 activation of imnet_k / imnet_v
    float* Hk;  // [nq*J][wk0]
    float* Hv;  // [nq*J][wv0]
    int* q_idx;
    int* k_idx;
};

struct LocalAttnP {
    const float* U;
    int ldu, D, Dv;   // D = 9C, Dv = 9C + Cn
    const int* q_idx;
    const int* k_idx;
    const float* wk; int ldwk;
    const float* wv; int ldwv;
    float* z; int ldz;
    int Q, J;
    float scale;
};

struct DecodeP {
    const float* h; int ldh, width;      // [nq][width] activations feeding the last Linear
    const float* w; int ldw;             // [3][ldw]
    const float* b;                      // [3]
    const float* x_lr;                   // [3][H][W] or null
    const float* coord;
    long q0;
    int nq, H, W;
    float* rgb;                          // [Q][3] (global query index)
};

// gemm_f32.hip
int gemm_f32(const float* A, int lda, const float* B, int ldb, bool b_kn, float* C, int ldc, const float* bias,
             int M, int N, int K, float alpha, int act, float slope, hipStream_t stream, const char* tag);
int gemm_f32_splitk(const float* A, int lda, const float* B, int ldb, bool b_kn, float* C, int ldc, const float* bias,
                    int M, int N, int K, float alpha, int act, float slope, float* partial, size_t partial_floats,
                    hipStream_t stream, const char* tag);
// The same two contractions with A = row_softmax(logits) formed in the operand staging: A holds the LOGITS, a_stats[row * a_stats_stride]
// = (row max x log2 e, 1 / sum of exp) from softmax_stats_rows; element (m, k) enters the MFMA as exp2(A[m][k] log2 e - max log2 e) / sum
// (one FMA, v_exp_f32, one multiply: the probabilities agree with softmax_rows' to fp32 rounding, not bitwise).  partial == nullptr:
// the plain kernel, else split-K.
int gemm_f32_softmax_a(const float* A, int lda, const float* a_stats2, int a_stats_stride, const float* B, int ldb, bool b_kn, float* C, int ldc,
                       int M, int N, int K, float* partial, size_t partial_floats, hipStream_t stream, const char* tag);
// the same contraction with one 192 x 256 workgroup tile per CU (gemm_big_f32.hip): [k][n] B, K a multiple of 16, >= 256 workgroup tiles
bool gemm_big_softmax_f32_ok(int lda, int ldb, int M, int N, int K, bool b_kn);
int gemm_big_softmax_f32(const float* A, int lda, const float* a_stats2, int a_stats_stride, const float* B, int ldb, float* C, int ldc,
                         int M, int N, int K, hipStream_t stream, const char* tag);
// gemm_small_f32.hip: few-MFLOP problems (no staging, K-sliced waves); gemm_small_ok tells whether a problem qualifies
bool gemm_small_ok(int M, int N, int K, int lda, int ldw);
int gemm_small_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, float* C2, int ldc2,
                   const float* res, int ldres, int M, int N, int K, int act, float slope, float alpha, hipStream_t s, const char* tag);
// dense_wino_f32.hip: fp32 dense layers of big maps in Winograd F(2x2, 3x3) form
int dense_wino_tiles(int H, int W);
int dense_layer_wino_f32(float* X, int ldx, int H, int W, int l, const float* frag_wino, const float* bias, int n_img, hipStream_t s);
// dense_wino4_f32.hip: the same layers in Winograd F(4x4, 3x3) form
int dense_wino4_tiles(int H, int W);
int dense_layer_wino4_f32(float* X, int ldx, int H, int W, int l, const float* frag_wino4, const float* bias, int n_img, hipStream_t s);
// nine 3x3 convolutions 64 -> 64 n_blk of the maps Pi[9][H*W][64] into out[(pix * 9 + o) * ldg + n] (logit table of the fused head)
int wino_table_f32(const float* Pi, int H, int W, const float* frag_wino, int n_blk, float* out, int ldg, hipStream_t s);
int wino4_table_f32(const float* Pi, int H, int W, const float* frag_wino4, int n_blk, float* out, int ldg, hipStream_t s);   // dense_wino4_f32.hip
// csa_scores_f32.hip: fp32 cs_attn correlation scores as a 3x3 diagonal box sum of the per-pixel correlation (Ch = 32)
bool csa_scores_box_ok(int Ch, int ldm, int ldr);
int csa_scores_box_f32(const float* M, int ldm, int Hp, int Wp, const float* R, int ldr, int Hl, int Wl, int Ch, float alpha, float floor_,
                       float* nrm, float* S, int ld_s, hipStream_t s);
// conv1x1_f32.hip: weights-resident fp32 1x1 convolution to 64 channels (the RDB local feature fusion on big maps)
bool conv1x1_resident_ok(long M, int N, int K, int ldx, int ldw);
int conv1x1_resident_f32(const float* X, int ldx, const float* W, int ldw, const float* bias, const float* res, int ldres, float* dst,
                         int ld_dst, float* dst2, int ld_dst2, long M, int K, hipStream_t s, const char* tag);
// conv_small_f32.hip: 3x3 convolutions of small maps in one launch (needs ciaosr_pack_fragments_f32 weights)
bool conv3x3_small_ok(int H, int W, int Cin, int Cout, int ld_src, int act);
int conv3x3_small(const float* src, int ld_src, int H, int W, int Cin, const float* frag, const float* bias, int Cout, float* dst,
                  int ld_dst, float* dst2, int ld_dst2, const float* res, int ld_res, int act, float alpha, hipStream_t s,
                  const char* tag);
// patch_ops.hip
int nchw_to_hwc(const float* src, float* dst, int C, int H, int W, int ld, hipStream_t s);
int hwc_to_nchw(const float* src, int ld, float* dst, int C, int H, int W, hipStream_t s);
int pad_reflect(const float* src, int ld_src, int H, int W, int C, float* dst, int Hp, int Wp, hipStream_t s);
int avgpool2(const float* src, int Hp, int Wp, int C, float* dst, hipStream_t s);
int patch_rows(const float* src, int ld_src, int Hs, int Ws, int Cs, int k, int stride, int pad, int OH, int OW,
               float* out, int ld_out, int normalize, float floor_, hipStream_t s, const char* tag);
int softmax_rows(float* S, long rows, int L, int ld, hipStream_t s);
int softmax_stats_rows(const float* S, long rows, int L, int ld, float* stats2 /* [rows] (max * log2 e, 1 / sum) */, hipStream_t s);
int fold(const float* O, int ldo, int Hp, int Wp, int C, float* Y, hipStream_t s);
int fold_s(const float* O, int ldo, int Hp, int Wp, int C, int scale, float* Y, hipStream_t s);
int downsample(const float* src, int Hp, int Wp, int C, int scale, float* dst, hipStream_t s);
int csa_gather_vprime(const float* Pc, int Hh, int Wh, int C, float* Vp, hipStream_t s);
int csa_gather_out(const float* Op, const float* Otop, const float* Oleft, const float* Otl, const float* bd, int H, int W,
                   int Hp, int Wp, int C, float* out, int ld_out, long ld_main, long ld_top, long ld_left, hipStream_t s);
int csa_gather_vprime_t_h16(const float* Pc, int Hh, int Wh, int C, unsigned short* VpT, int ldt, bool f16, hipStream_t s);
// head_ops.hip
int head_indices(const float* coord, const float* cell, long q0, int nq, int chunk, int H, int W, int local_size,
                 int* q_idx, int* k_idx, float* rel, hipStream_t s);
int head_rows(const HeadRowsP& p, hipStream_t s);
int qk_rows(const float* U, int ldu, int D, int H, int W, long row0, int nrows, const float* bias_out, float* A, float* G,
            int ldg, int rows_h16 /* 0 fp32 rows, 1 bf16, 2 half, 3 no rows (the bias term only) */, hipStream_t s);
int qk_maps(const float* F, int ldf, int C, int H, int W, float* Pi, hipStream_t s);
int transpose_cast_h16(const float* W, int ld, int K, int N, unsigned short* out, bool f16, hipStream_t s);
int local_attention(const LocalAttnP& p, hipStream_t s);
int decode_residual(const DecodeP& p, hipStream_t s);

// head_fused.hip
struct FusedChain {
    const float* table;
    const float* tail;
    int ld_tail;
    const void* frag_hidden[CIAOSR_MAX_LAYERS];
    const void* frag_hidden_lo[CIAOSR_MAX_LAYERS];   // bf16 kernels: low half of the hi + lo weight pair, or null
    const float* bias_hidden[CIAOSR_MAX_LAYERS];
    int n_hidden;
    const void* frag_out;
    const void* frag_out_lo;
    const float* bias_out;
    int n_out;
};
struct FusedKVP {
    const float* coord;
    const float* cell;
    long q0;
    int nq, chunk, H, W;
    const float* U;
    int ldu, D, Dv;
    unsigned u_bytes;
    FusedChain k, v;
    float softmax_scale;
    float* Z;
    int ldz;
    // optional logit table (head.hip): G[(qpix*9 + (oy+1)*3 + (ox+1))][0..255] = W5k^T (q*key), [256] = b5k.(q*key)
    const float* G;
    int ldg;
    unsigned g_bytes;
    int rows_per_wg;   // fp32 kernel: 32 (0 = default) or 64
    const int* gate;   // 16-bit 128-row kernel as the fallback of the chained one: run only when *gate != 0 (null = always)
};
struct FusedQP {
    const float* Z; int ldz, Dv;
    const void* frag_in;
    const void* frag_in_lo;                          // bf16 kernel: low half of the hi + lo weight pair, or null
    const float* bias_in;
    int nj_in;
    const void* frag_hidden[CIAOSR_MAX_LAYERS];
    const void* frag_hidden_lo[CIAOSR_MAX_LAYERS];
    const float* bias_hidden[CIAOSR_MAX_LAYERS];
    int n_hidden;
    const float* w_last; int ld_last;
    const float* b_last;
    const float* x_lr;
    const float* coord;
    long q0;
    int nq, H, W;
    float* rgb;
    int rows_per_wg;   // fp32 kernel: 32 (0 = default) or 64
};
int head_kv_fused(const FusedKVP& p, hipStream_t s);
int head_decode_fused(const FusedQP& p, hipStream_t s);

// The 16-bit translation units (head_fused_h16.hip, gemm_h16.hip, dense_h16.hip) are compiled once per element type
// (h16_util.h): the same functions exist in ciaosr::b16 (bf16) and ciaosr::f16 (IEEE half).
#define CIAOSR_H16_DECLS                                                                                                              \
    int gemm_h16_nt(const unsigned short* A, int lda, const unsigned short* B, int ldb, void* C, int ldc, bool c_bf16, int M, int N,  \
                    int K, float alpha, hipStream_t s, const char* tag);                                                              \
    int cast_rows_h16(const float* src, int ld_src, unsigned short* dst, int ld_dst, long rows, int cols, hipStream_t s);             \
    int linear_h16(const unsigned short* A, int lda, const unsigned short* W16, int ldw, const float* bias, bool relu, void* C, int ldc,      \
                   bool c_16bit, int M, int N, int K, hipStream_t s, const char* tag);                                                      \
    int softmax_rows_h16(const float* S, long rows, int L, int ld, unsigned short* P, int ldp, hipStream_t s);                        \
    int head_kv_fused_h16(const FusedKVP& p, hipStream_t s);                                                                          \
    int head_decode_fused_h16(const FusedQP& p, hipStream_t s);                                                                       \
    bool head_chain_ok(const ciaosr_head_weights_t* w);                                                                               \
    size_t head_chain_bytes(const ciaosr_head_weights_t* w, int pairs);                                                               \
    int pack_head_chain(const ciaosr_head_weights_t* w, int pairs, void* out, hipStream_t s);                                         \
    bool head_decode_chain_ok(const ciaosr_head_weights_t* w);                                                                        \
    size_t head_kv_chain_bytes(const ciaosr_head_weights_t* w, int pairs);                                                            \
    int head_decode_chain_h16(const FusedQP& qp, const ciaosr_head_weights_t* w, const void* blob, int pairs, hipStream_t s);         \
    int head_kv_chain_h16(const FusedKVP& kp, const ciaosr_head_weights_t* w, const void* blob, int pairs, int grid_w, int* flag,     \
                          hipStream_t s);                                                                                             \
    int pack_fragments_h16(const float* W, int ld, int N, int K, void* P, hipStream_t s, int residual, void* P_lo);                   \
    namespace wide {                                                                                                                  \
    int head_kv_fused_wide(const FusedKVP& p, int mode, hipStream_t s);                                                               \
    int head_decode_fused_wide(const FusedQP& p, int mode, hipStream_t s);                                                            \
    }                                                                                                                                 \
    int dense_h16_tiles(int H, int W);                                                                                                \
    int cast_group_h16(const float* X, int ldx, unsigned short* Xb, int ldxb, int col, long HW, hipStream_t s);                       \
    int dense_layer_h16(float* X, int ldx, unsigned short* Xb, int ldxb, int H, int W, int l, const void* frag16,                     \
                        const void* frag16_lo, const float* bias, int n_img, hipStream_t s, int route);                               \
    int conv1x1_h16(const unsigned short* A, int lda, const unsigned short* W16, int ldw, const float* bias, const float* res,        \
                    int ldres, float* out, int ldo, float* out2, int ldo2, unsigned short* out16, int ldo16, int M, int N, int K,     \
                    hipStream_t s, const char* tag);                                                                                  \
    int cast_many_h16(const float* const* src, int n, int rows, int cols, unsigned short* dst, hipStream_t s);                       \
    size_t softmax_gemm_scratch_floats(long M, int N);                                                                                \
    int softmax_gemm_h16_nt(const unsigned short* A, int lda, const unsigned short* B, int ldb, unsigned short* P, int ldp, int M,    \
                            int N, int K, float alpha, float* scratch, size_t scratch_floats, hipStream_t s, const char* tag);        \
    int rows_to_f32_h16(const unsigned short* src, long ld_src, long row0, long row_stride, int nrows, int cols, float* dst,          \
                        int ld_dst, hipStream_t s);
namespace b16 { CIAOSR_H16_DECLS }
namespace f16 { CIAOSR_H16_DECLS }
#undef CIAOSR_H16_DECLS

// (head_fused_wide_h16.hip, declared per element type above: the fused head with one wide workgroup per CU.  mode 0 = f16 (256 rows), 1 = f16-pairs
// (256 rows, weights as hi + lo pairs), 2 = f16x3 / bf16x3 (128 rows, weights AND activations as pairs: three MFMAs per product, Z in fp32); the
// bf16 build carries mode 2 only)

// precision of an entry point: the suffix of its name
enum Prec { kF32 = 0, kBF16 = 1, kF16 = 2 };
struct H16Ops {
    decltype(&b16::gemm_h16_nt) gemm_nt;
    decltype(&b16::cast_rows_h16) cast_rows;
    decltype(&b16::softmax_rows_h16) softmax_rows;
    decltype(&b16::head_kv_fused_h16) head_kv_fused;
    decltype(&b16::head_decode_fused_h16) head_decode_fused;
    decltype(&b16::cast_group_h16) cast_group;
    decltype(&b16::dense_layer_h16) dense_layer;
    decltype(&b16::conv1x1_h16) conv1x1;
    decltype(&b16::cast_many_h16) cast_many;
    decltype(&b16::rows_to_f32_h16) rows_to_f32;
    decltype(&b16::softmax_gemm_h16_nt) softmax_gemm_nt;
    decltype(&b16::softmax_gemm_scratch_floats) softmax_gemm_scratch;
    decltype(&b16::head_chain_ok) head_chain_ok;
    decltype(&b16::head_kv_chain_h16) head_kv_chain;
    decltype(&b16::head_decode_chain_ok) head_decode_chain_ok;
    decltype(&b16::head_kv_chain_bytes) head_kv_chain_bytes;
    decltype(&b16::head_decode_chain_h16) head_decode_chain;
};
inline const H16Ops& h16_ops(Prec prec) {
    static const H16Ops kB = {b16::gemm_h16_nt, b16::cast_rows_h16, b16::softmax_rows_h16, b16::head_kv_fused_h16,
                              b16::head_decode_fused_h16, b16::cast_group_h16, b16::dense_layer_h16, b16::conv1x1_h16,
                              b16::cast_many_h16, b16::rows_to_f32_h16, b16::softmax_gemm_h16_nt,
                              b16::softmax_gemm_scratch_floats, b16::head_chain_ok, b16::head_kv_chain_h16,
                              b16::head_decode_chain_ok, b16::head_kv_chain_bytes, b16::head_decode_chain_h16};
    static const H16Ops kH = {f16::gemm_h16_nt, f16::cast_rows_h16, f16::softmax_rows_h16, f16::head_kv_fused_h16,
                              f16::head_decode_fused_h16, f16::cast_group_h16, f16::dense_layer_h16, f16::conv1x1_h16,
                              f16::cast_many_h16, f16::rows_to_f32_h16, f16::softmax_gemm_h16_nt,
                              f16::softmax_gemm_scratch_floats, f16::head_chain_ok, f16::head_kv_chain_h16,
                              f16::head_decode_chain_ok, f16::head_kv_chain_bytes, f16::head_decode_chain_h16};
    return prec == kF16 ? kH : kB;
}

// bump allocator over the caller-provided workspace (256-byte aligned carve-outs)
struct Arena {
    char* base;
    size_t size, off;
    bool ok;
    Arena(void* p, size_t n) : base((char*)p), size(n), off(0), ok(true) {}
    template <typename T>
    T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* r = reinterpret_cast<T*>(base + off);
        off += count * sizeof(T);
        if (off > size) ok = false;
        return r;
    }
};

static inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace ciaosr
