// Microbenchmark: how VALU issue and the fp32 MFMA share a SIMD on gfx950.
//   hipcc -O3 --offload-arch=gfx950 mfma_valu.hip -o mfma_valu && ./mfma_valu
// One workgroup per CU slot measured (grid = 256), W waves per SIMD (block = 256 W), each wave runs R rounds of
// [4 independent v_mfma_f32_32x32x2_f32 + K independent VALU per MFMA (+ L ds_read_b128 per 4 MFMAs)]; prints cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: K VALU behind every MFMA; 1: the 4 K VALU of a round in one cluster behind its 4 MFMAs; 2: as 0 with v_pk_fma_f32 (K/2 instructions);
// 3: 16 MFMAs, then their 16 K VALU in one cluster (the shape hipcc gives the Winograd step)
template <int K, int L, int MODE, int H16 = 0>
__global__ __launch_bounds__(1024) void bench(float* out, unsigned long long* cyc, int rounds) {
    __shared__ float4 lds[4096];
    const int t = threadIdx.x;
    for (int i = t; i < 4096; i += blockDim.x) lds[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = t * 0.001f, b = 1.0f + t;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = t * 0.5f + i;
    float4 dd[4];
    for (int i = 0; i < 4; ++i) dd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b - i); }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 px[4];
    for (int i = 0; i < 4; ++i) px[i] = f32x2{x[i], x[i + 4]};
    const f32x2 pa = {a, a};
    const float4* lp = lds + (t * 17 & 2047);
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 3) {
            if ((r & 3) == 3) {
#pragma unroll
                for (int k = 0; k < 16 * K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[k & 7]) : "v"(a));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (H16) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(j * K + k) & 7]) : "v"(a));
            } else if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < K / 2; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(px[(j * K + k) & 3]) : "v"(pa));
            } else if (MODE == 1 && j == 3) {
#pragma unroll
                for (int k = 0; k < 4 * K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[k & 7]) : "v"(a));
            }
            if (j < L) dd[j] = lp[j * 64 + (r & 1) * 256];          // immediate offsets: no address VALU
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (int i = 0; i < 4; ++i) { x[i] += px[i].x; x[i + 4] += px[i].y; }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += dd[i].x + dd[i].y + dd[i].z + dd[i].w;
    out[blockIdx.x * blockDim.x + t] = s;
    if ((t & 63) == 0) atomicMax(&cyc[blockIdx.x], t1 - t0);
}

template <int K, int L, int MODE = 0, int H16 = 0>
static void run(int waves_per_simd, int rounds = 2000) {
    float* out; unsigned long long* cyc;
    const int grid = 256, block = 256 * waves_per_simd;
    hipMalloc(&out, sizeof(float) * grid * block);
    hipMalloc(&cyc, sizeof(unsigned long long) * grid);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, sizeof(unsigned long long) * grid);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((bench<K, L, MODE, H16>), dim3(grid), dim3(block), 0, 0, out, cyc, rounds);
        hipEventRecord(e1, 0);
    }
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < grid; ++i) avg += (double)h[i];
    avg /= grid;
    const double per_mfma = avg / (rounds * 4.0 * waves_per_simd);
    static const char* mode[] = {"interleaved", "cluster per 4 MFMAs", "packed (K/2 v_pk_fma)", "cluster per 16 MFMAs"};
    printf("%s waves/SIMD %d  VALU per MFMA %2d (%s)  ds_read_b128 per 4 MFMA %d : %.1f cycles per MFMA on the SIMD (peak: 64 f32 / 32 f16, slowest wave); kernel %.3f ms by HIP events = %.2f counter ticks per ns\n", H16 ? "f16 32x32x16" : "f32 32x32x2 ", waves_per_simd, K, mode[MODE], L, per_mfma, ms, avg / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
    if (argc > 1) {      // sustained-load check: pure MFMA streams of growing length (does the clock hold?)
        for (int r : {2000, 20000, 100000, 400000}) { run<0, 0>(4, r); run<0, 0, 0, 1>(4, r); }
        return 0;
    }
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 0>(w); run<1, 0>(w); run<2, 0>(w); run<3, 0>(w); run<4, 0>(w); run<6, 0>(w); run<8, 0>(w);
        run<2, 0, 1>(w); run<4, 0, 1>(w); run<2, 0, 3>(w); run<4, 0, 3>(w); run<4, 0, 2>(w); run<8, 0, 2>(w);
        run<0, 1>(w); run<0, 2>(w); run<0, 4>(w); run<2, 2>(w); run<4, 2, 2>(w);
        run<0, 0, 0, 1>(w); run<1, 0, 0, 1>(w); run<2, 0, 0, 1>(w); run<4, 0, 0, 1>(w); run<8, 0, 0, 1>(w); run<2, 0, 1, 1>(w); run<4, 0, 1, 1>(w); run<4, 0, 3, 1>(w); run<4, 0, 2, 1>(w); run<0, 2, 0, 1>(w); run<0, 4, 0, 1>(w);
    }
    return 0;
}
