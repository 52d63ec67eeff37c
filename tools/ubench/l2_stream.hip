// How fast can every CU stream the SAME weight fragments out of L2?  (The 16-bit fused head kernels all sit at ~14.7 B/clk per CU of
// weight stream -- narrow f16 0.46 x 32, f16x3 0.70 x 21, wide f16 128 KB per 8.9k cycles -- far below the L2's nominal bandwidth.)
// One workgroup of 8 waves per CU, 256 CUs x ROUNDS; a wave walks `nfrag` 1-KB fragments (64 lanes x 16 B) of its own 1/8 of a buffer
// of `bytes` bytes, `depth` loads in flight, and folds them into a checksum.
//   variant 0: every workgroup walks the fragments in the same order (what the head kernels do)
//   variant 1: workgroup b starts at fragment (b * 7) % nfrag (same bytes, de-correlated in time)
//   variant 2: every workgroup has its own copy of the buffer (no sharing at all: the L2 / HBM-side limit)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/l2_stream.hip -o tools/ubench/l2_stream && tools/ubench/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(512) void stream(const uint4* __restrict__ buf, int nfrag, int variant, long copy_stride, int reps, unsigned* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint4* base = buf + (variant == 2 ? (long)(blockIdx.x % 256) * copy_stride : 0) + (long)w * nfrag * 64 + lane;
    const int rot = variant == 1 ? (int)((blockIdx.x * 7u) % (unsigned)nfrag) : 0;
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        uint4 v[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) { int f = i + rot; if (f >= nfrag) f -= nfrag; v[i] = base[(long)f * 64]; }
        for (int s = 0; s < nfrag; s += DEPTH) {
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
                int f = s + DEPTH + i; if (f >= nfrag) f -= nfrag;      // wraps into the next rep's first fragments
                f += rot; if (f >= nfrag) f -= nfrag;
                v[i] = base[(long)f * 64];
            }
        }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) acc += v[i].x;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const int nfrag = 16 * 8;                     // per wave: 128 KB (a 256-wide layer's hi fragments are 16 per wave; x8 = a whole chain)
    const size_t bytes = (size_t)8 * nfrag * 1024;   // 1 MB per copy
    uint4* d; unsigned* o;
    hipMalloc(&d, bytes * 256); hipMalloc(&o, 64);
    hipMemset(d, 1, bytes * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 3; ++variant)
        for (int depth : {2, 4, 8}) {
            const int reps = 8, grid = 256 * 8;
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                hipEventRecord(e0);
                if (depth == 2) hipLaunchKernelGGL(stream<2>, dim3(grid), dim3(512), 0, 0, d, nfrag, variant, (long)(bytes / 16), reps, o);
                if (depth == 4) hipLaunchKernelGGL(stream<4>, dim3(grid), dim3(512), 0, 0, d, nfrag, variant, (long)(bytes / 16), reps, o);
                if (depth == 8) hipLaunchKernelGGL(stream<8>, dim3(grid), dim3(512), 0, 0, d, nfrag, variant, (long)(bytes / 16), reps, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double total = (double)grid * reps * bytes;
            printf("variant %d depth %d: %.3f ms  %.2f TB/s chip = %.1f B/clk/CU at 2.4 GHz\n", variant, depth, best, total / best / 1e9,
                   total / (best * 1e-3) / 256 / 2.4e9);
        }
    return 0;
}
