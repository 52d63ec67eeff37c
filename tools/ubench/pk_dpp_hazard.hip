// Root-cause probe for the SLP question: is there a read-after-write hazard between a PACKED fp32 VALU result (v_pk_add_f32 /
// v_pk_fma_f32, what hipcc's SLP vectoriser emits) and a DPP read of it (v_mov_b32_dpp quad_perm, what quad_xor1 lowers to) that the
// instruction stream hipcc generates does not cover?  The whole sequence sits in ONE asm block with fixed registers, so the distance
// between producer and DPP consumer is exactly what the FILL variant says:
//     v20, v21 <- sentinel; long wait; v_pk_add_f32 v[20:21], x, y; FILL; v_mov_b32_dpp v22, v20 quad_perm:[1,0,3,2]; ...
// A lane whose DPP read returns the sentinel (or anything but the neighbour's x + y) read STALE data.  Counts are per 16-lane row.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/pk_dpp_hazard.hip -o tools/ubench/pk_dpp_hazard && tools/ubench/pk_dpp_hazard
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define SEQ(PRODUCER, FILL)                                                                                       \
    asm volatile("v_mov_b32 v20, %[s]\n\tv_mov_b32 v21, %[s]\n\ts_nop 7\n\ts_nop 7\n\t" PRODUCER "\n\t" FILL      \
                 "v_mov_b32_dpp v22, v20 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
                 "v_mov_b32_dpp v23, v21 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
                 "s_nop 7\n\ts_nop 7\n\tv_mov_b32 %[d0], v22\n\tv_mov_b32 %[d1], v23\n\t"                          \
                 : [d0] "=v"(d0), [d1] "=v"(d1) : [x] "v"(x), [y] "v"(y), [xs] "v"(xs), [ys] "v"(ys), [s] "v"(sentinel) : "v20", "v21", "v22", "v23")

template <int VARIANT>
__global__ void probe(unsigned* bad /* [4 rows] */, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned mism[2] = {0, 0};
    for (int it = 0; it < iters; ++it) {
        const f32x2 x = {(float)(lane * 3 + it), (float)(lane * 5 - it)};
        const f32x2 y = {(float)(it & 7) + 0.5f, 1.25f};
        const float sentinel = -12345.f, xs = x.x, ys = y.x;
        float d0, d1;
        if (VARIANT == 0) SEQ("v_pk_add_f32 v[20:21], %[x], %[y]", "");
        if (VARIANT == 1) SEQ("v_pk_add_f32 v[20:21], %[x], %[y]", "v_nop\n\t");
        if (VARIANT == 2) SEQ("v_pk_add_f32 v[20:21], %[x], %[y]", "v_nop\n\tv_nop\n\t");
        if (VARIANT == 3) SEQ("v_pk_add_f32 v[20:21], %[x], %[y]", "v_nop\n\tv_nop\n\tv_nop\n\t");
        if (VARIANT == 4) SEQ("v_pk_add_f32 v[20:21], %[x], %[y]", "v_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\t");
        if (VARIANT == 5) SEQ("v_pk_add_f32 v[20:21], %[x], %[y]", "s_nop 1\n\t");
        if (VARIANT == 6) SEQ("v_pk_add_f32 v[20:21], %[x], %[y]", "s_nop 4\n\t");
        if (VARIANT == 7) SEQ("v_add_f32 v20, %[xs], %[ys]\n\tv_add_f32 v21, %[xs], %[ys]", "");       // plain VALU producer, no fill (reference)
        if (VARIANT == 8) SEQ("v_pk_fma_f32 v[20:21], %[x], %[y], %[y]", "");
        if (VARIANT == 9) SEQ("v_pk_add_f32 v[20:21], %[x], %[y]", "s_mov_b32 s20, 0\n\ts_mov_b32 s21, 0\n\tv_nop\n\tv_nop\n\t");   // SALU + 2 VALU (the kernel's distance)
        // expected: neighbour lane's sum
        const int nb = lane ^ 1;
        const float e0 = VARIANT == 7 ? (float)(nb * 3 + it) + ((float)(it & 7) + 0.5f)
                                      : (VARIANT == 8 ? (float)(nb * 3 + it) * ((float)(it & 7) + 0.5f) + ((float)(it & 7) + 0.5f)
                                                      : (float)(nb * 3 + it) + ((float)(it & 7) + 0.5f));
        const float e1 = VARIANT == 7 ? (float)(nb * 3 + it) + ((float)(it & 7) + 0.5f)     // v21 = x.x + y.x too in the scalar reference
                                      : (VARIANT == 8 ? (float)(nb * 5 - it) * 1.25f + 1.25f : (float)(nb * 5 - it) + 1.25f);
        mism[0] += d0 != e0;
        mism[1] += d1 != e1;
    }
    if (mism[0] | mism[1]) atomicAdd(&bad[lane >> 4], mism[0] + mism[1]);
}

template <int V>
static void run(const char* name, unsigned* d) {
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(probe<V>, dim3(1024), dim3(256), 0, 0, d, 2000);
    unsigned h[4];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-58s stale DPP reads by 16-lane row: %u %u %u %u\n", name, h[0], h[1], h[2], h[3]);
}

int main() {
    unsigned* d;
    hipMalloc(&d, 16);
    run<7>("v_add_f32 x2 -> dpp, no fill (reference)", d);
    run<0>("v_pk_add_f32 -> dpp, no fill", d);
    run<1>("v_pk_add_f32 -> 1 v_nop -> dpp", d);
    run<2>("v_pk_add_f32 -> 2 v_nop -> dpp", d);
    run<3>("v_pk_add_f32 -> 3 v_nop -> dpp", d);
    run<4>("v_pk_add_f32 -> 4 v_nop -> dpp", d);
    run<5>("v_pk_add_f32 -> s_nop 1 -> dpp", d);
    run<6>("v_pk_add_f32 -> s_nop 4 -> dpp", d);
    run<8>("v_pk_fma_f32 -> dpp, no fill", d);
    run<9>("v_pk_add_f32 -> 2 SALU + 2 v_nop -> dpp", d);
    return 0;
}
