// Model-load path (round 5): every MFMA fragment form of a 3x3 convolution weight in ONE launch -- the direct [N][9K] matrix, the
// Winograd F(2x2, 3x3) form U = G g G^T (16 positions) and the F(4x4, 3x3) form (36 positions), each position its own [N][K] matrix in
// ciaosr_pack_fragments_f32 order -- with the transform evaluated in fp64 on the device and rounded once.  Replaces, per dense layer of
// the RDN trunk, two fp64 einsum calls (rocBLAS / Tensile DGEMMs on the device) and 53 pack_fragments launches: 6 978 launches per model
// in round 4, ~300 now; the library no longer depends on a vendor GEMM at load time.
// Reference: the weights are mmedit RDN's DenseLayer convolutions (ciaosr_net.py:321-342) and imnet_k's output layer read as nine 3x3
// convolutions (head.hip, logit table).
#include "ops.h"

namespace ciaosr {

struct PackConvP {
    const float* w;             // element (o, a, b, c) at w[o * so + a * sa + b * sb + c * sc]
    long so, sa, sb, sc;
    int N, K;                   // output channels, input channels (K a multiple of 4)
    float* f_direct;            // [nt][ceil(9K / 8)][64][4] of the [N][(3a + b) K + c] matrix, or null
    float* f_w2;                // 16 x [nt][ceil(K / 8)][64][4], or null
    float* f_w4;                // 36 x ..., or null
};

__device__ __forceinline__ double g2(int i, int a) {      // G of F(2, 3)
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    return G[i][a];
}
__device__ __forceinline__ double g4(int i, int a) {      // G of F(4, 3)
    const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 24, 1.0 / 12, 1.0 / 6},
                            {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1.0}};
    return G[i][a];
}

__global__ void pack_conv3x3_kernel(PackConvP p) {
    const int nt = (p.N + 31) / 32, nj = (p.K + 7) / 8, njd = (9 * p.K + 7) / 8;
    const long per_pos = (long)nt * nj * 64;
    const long n_direct = p.f_direct ? (long)nt * njd * 64 : 0, n_w2 = p.f_w2 ? 16 * per_pos : 0, n_w4 = p.f_w4 ? 36 * per_pos : 0;
    const long total = n_direct + n_w2 + n_w4;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < n_direct) {
            const int lane = (int)(idx & 63);
            const long t = idx >> 6;
            const int j = (int)(t % njd), tile = (int)(t / njd);
            const int n = tile * 32 + (lane & 31), k = 8 * j + 4 * (lane >> 5);
            if (n < p.N && k < 9 * p.K) {
                const int tap = k / p.K, c = k - tap * p.K;           // K % 4 == 0: the four columns share a tap
                const float* r = p.w + n * p.so + (tap / 3) * p.sa + (tap % 3) * p.sb + c * p.sc;
                v = make_float4(r[0], r[p.sc], r[2 * p.sc], r[3 * p.sc]);
            }
            reinterpret_cast<float4*>(p.f_direct)[idx] = v;
            continue;
        }
        const bool four = idx >= n_direct + n_w2;
        const long q = idx - n_direct - (four ? n_w2 : 0);
        const int pos = (int)(q / per_pos);
        const long r_ = q - pos * per_pos;
        const int lane = (int)(r_ & 63);
        const long t = r_ >> 6;
        const int j = (int)(t % nj), tile = (int)(t / nj);
        const int n = tile * 32 + (lane & 31), k = 8 * j + 4 * (lane >> 5);
        const int side = four ? 6 : 4, pi = pos / side, pj = pos % side;
        if (n < p.N && k < p.K) {
            double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double ga = four ? g4(pi, a) : g2(pi, a);
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const double gg = ga * (four ? g4(pj, b) : g2(pj, b));
                    const float* r = p.w + n * p.so + a * p.sa + b * p.sb + k * p.sc;
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] += gg * (double)r[e * p.sc];
                }
            }
            v = make_float4((float)s[0], (float)s[1], (float)s[2], (float)s[3]);
        }
        reinterpret_cast<float4*>(four ? p.f_w4 : p.f_w2)[q] = v;
    }
}

}  // namespace ciaosr

using namespace ciaosr;

extern "C" int ciaosr_pack_conv3x3_f32(const float* w, size_t stride_o, size_t stride_a, size_t stride_b, size_t stride_c, int N, int K,
                                       float* frag_direct, float* frag_wino2, float* frag_wino4, void* stream) {
    CIAOSR_CHECK_ARG(w && N > 0 && K > 0 && (K & 3) == 0 && (frag_direct || frag_wino2 || frag_wino4));
    PackConvP p{w, (long)stride_o, (long)stride_a, (long)stride_b, (long)stride_c, N, K, frag_direct, frag_wino2, frag_wino4};
    const long nt = (N + 31) / 32;
    const long total = (frag_direct ? nt * ((9 * K + 7) / 8) * 64 : 0) + (long)((frag_wino2 ? 16 : 0) + (frag_wino4 ? 36 : 0)) * nt * ((K + 7) / 8) * 64;
    const int grid = (int)((total + 255) / 256);
    ProfScope prof("pack_conv3x3", (hipStream_t)stream);
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(grid > 8192 ? 8192 : grid), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("pack_conv3x3");
}
