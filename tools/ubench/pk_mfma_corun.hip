// Root-cause probe for the SLP question, part 2.  Finding so far (tools/slp_bisect.py): with the SLP vectoriser on, the 16-bit decode
// kernel's last-Linear loop (packed fp32 VALU: v_pk_mul_f32 / v_pk_fma_f32 / v_pk_mov_b32 with op_sel) gives run-to-run different
// results in lanes 48-63 -- but ONLY when two workgroups share a CU (the same code with one workgroup per CU is deterministic).  So:
// do packed fp32 VALU instructions of one wave return wrong results while ANOTHER wave of the same SIMD issues MFMAs?
// 512-thread workgroup = two waves per SIMD.  Waves 0-3 ("valu") evaluate a packed instruction and the same thing with scalar
// instructions on lane-dependent data, `iters` times, and count mismatches per 16-lane row; waves 4-7 ("partner") loop on MFMAs (or
// idle, or plain VALU) until the valu waves are done.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/pk_mfma_corun.hip -o tools/ubench/pk_mfma_corun && tools/ubench/pk_mfma_corun
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// OP: 0 v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]   1 v_pk_fma_f32   2 v_pk_fma_f32 op_sel_hi:[1,0,1]   3 v_pk_mov_b32 op_sel:[1,0]
//     4 v_pk_add_f32                                 5 a chain of all of them (the kernel's mix)
template <int OP>
__device__ __forceinline__ void both(f32x2 a, f32x2 b, f32x2 c, f32x2& pk, f32x2& sc) {
    if constexpr (OP == 0) {
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(pk) : "v"(a), "v"(b));
        sc = f32x2{a.x * b.y, a.y * b.x};
    } else if constexpr (OP == 1) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pk) : "v"(a), "v"(b), "v"(c));
        sc = f32x2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)};
    } else if constexpr (OP == 2) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(pk) : "v"(a), "v"(b), "v"(c));
        sc = f32x2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.x, c.y)};
    } else if constexpr (OP == 3) {
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(pk) : "v"(a), "v"(b));
        sc = f32x2{a.y, b.x};
    } else if constexpr (OP == 4) {
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
        sc = f32x2{a.x + b.x, a.y + b.y};
    } else if constexpr (OP == 6) {
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
        sc = f32x2{a.x * b.x, a.y * b.y};
    } else if constexpr (OP == 7) {
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pk) : "v"(a), "v"(b));       // b.x for both halves
        sc = f32x2{a.x * b.x, a.y * b.x};
    } else if constexpr (OP == 8) {
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(pk) : "v"(a), "v"(b));   // both operands swapped
        sc = f32x2{a.y * b.y, a.x * b.x};
    } else if constexpr (OP == 9) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(pk) : "v"(a), "v"(b), "v"(c));   // the same cross selection on an fma
        sc = f32x2{__builtin_fmaf(a.x, b.y, c.x), __builtin_fmaf(a.y, b.x, c.y)};
    } else if constexpr (OP == 10) {
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(pk) : "v"(a), "v"(b));
        sc = f32x2{a.x + b.y, a.y + b.x};
    } else {
        f32x2 t0, t1, t2;
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t0) : "v"(a), "v"(b));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t1) : "v"(t0), "v"(c));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t2) : "v"(a), "v"(b), "v"(t1));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pk) : "v"(t2), "v"(c));
        const f32x2 s0 = {a.y, b.x};
        const f32x2 s1 = {s0.x * c.y, s0.y * c.x};
        const f32x2 s2 = {__builtin_fmaf(a.x, b.x, s1.x), __builtin_fmaf(a.y, b.x, s1.y)};
        sc = f32x2{s2.x + c.x, s2.y + c.y};
    }
}

// PARTNER: 0 idle (exits at once)   1 v_mfma_f32_32x32x16_f16   2 v_mfma_f32_32x32x2_f32   3 plain v_fma_f32   4 v_mfma_f32_32x32x16_bf16
//          5 v_mfma_f32_16x16x32_f16
template <int OP, int PARTNER>
__global__ __launch_bounds__(512) void corun(unsigned* bad, int iters, float* sink) {
    __shared__ int done;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (w < 4) {
        unsigned mism = 0;
        f32x2 a = {1.0f + lane * 0.37f, 2.0f - lane * 0.11f}, b = {0.5f + lane * 0.03f, -1.5f + lane * 0.07f}, c = {3.0f, lane * 0.5f};
        for (int it = 0; it < iters; ++it) {
            f32x2 pk, sc;
            both<OP>(a, b, c, pk, sc);
            mism += (__float_as_uint(pk.x) != __float_as_uint(sc.x)) + (__float_as_uint(pk.y) != __float_as_uint(sc.y));
            a.x += 0.25f; b.y -= 0.125f; c.x = c.x * 0.5f + 1.0f;
        }
        if (mism) atomicAdd(&bad[lane >> 4], mism);
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) atomicAdd(&done, 1);
    } else {
        if (PARTNER == 0) return;
        f32x16 acc = {};
        f32x4 acc4 = {};
        f16x8 ha, hb;
        bf16x8 ba, bb;
        for (int i = 0; i < 8; ++i) {
            ha[i] = (_Float16)(0.01f * lane + i); hb[i] = (_Float16)(0.5f - 0.001f * lane);
            ba[i] = (__bf16)(0.01f * lane + i); bb[i] = (__bf16)(0.5f - 0.001f * lane);
        }
        float va = lane * 0.1f, vb = 1.0001f;
        for (int guard = 0; guard < 200000 && __hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4; ++guard) {   // bounded: never hangs
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (PARTNER == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
                if (PARTNER == 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, acc, 0, 0, 0);
                if (PARTNER == 3) va = __builtin_fmaf(va, vb, 0.5f);
                if (PARTNER == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc, 0, 0, 0);
                if (PARTNER == 5) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4, 0, 0, 0);
            }
        }
        if (acc[0] + va + acc4[0] == 123.456f) sink[0] = acc[3];
    }
}

template <int OP, int PARTNER>
static void run(const char* name, unsigned* d, float* sink) {
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL((corun<OP, PARTNER>), dim3(1024), dim3(512), 0, 0, d, 4000, sink);
    hipDeviceSynchronize();
    unsigned h[4];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-72s mismatches by 16-lane row: %u %u %u %u\n", name, h[0], h[1], h[2], h[3]);
    fflush(stdout);
}

int main() {
    unsigned* d; float* sink;
    hipMalloc(&d, 16); hipMalloc(&sink, 64);
    run<5, 0>("pk chain (mov/mul/fma/add with op_sel), partner idle", d, sink);
    run<5, 3>("pk chain, partner plain VALU", d, sink);
    run<5, 1>("pk chain, partner v_mfma_f32_32x32x16_f16", d, sink);
    run<5, 2>("pk chain, partner v_mfma_f32_32x32x2_f32", d, sink);
    run<0, 1>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0], partner f16 MFMA", d, sink);
    run<1, 1>("v_pk_fma_f32, partner f16 MFMA", d, sink);
    run<2, 1>("v_pk_fma_f32 op_sel_hi:[1,0,1], partner f16 MFMA", d, sink);
    run<3, 1>("v_pk_mov_b32 op_sel:[1,0], partner f16 MFMA", d, sink);
    run<4, 1>("v_pk_add_f32, partner f16 MFMA", d, sink);
    run<6, 1>("v_pk_mul_f32 (no op_sel), partner f16 MFMA", d, sink);
    run<7, 1>("v_pk_mul_f32 op_sel_hi:[1,0], partner f16 MFMA", d, sink);
    run<8, 1>("v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,0], partner f16 MFMA", d, sink);
    run<9, 1>("v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1], partner f16 MFMA", d, sink);
    run<10, 1>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0], partner f16 MFMA", d, sink);
    run<0, 4>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0], partner bf16 MFMA", d, sink);
    run<0, 5>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0], partner 16x16x32 f16 MFMA", d, sink);
    run<0, 2>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0], partner fp32 MFMA", d, sink);
    run<0, 3>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0], partner plain VALU", d, sink);
    run<0, 0>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0], partner idle", d, sink);
    run<6, 2>("v_pk_mul_f32 (no op_sel), partner fp32 MFMA", d, sink);
    return 0;
}
