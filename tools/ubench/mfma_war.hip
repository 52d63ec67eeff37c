// Microbenchmark (round 4): is a VALU write to the SrcA / SrcB register of an fp32 MFMA that was ISSUED a few cycles earlier safe?
// Found while scheduling dense_wino4_f32.hip: the T phase's first VALU instructions reuse the B-operand registers of the M phase's
// last MFMAs; with inline-asm MFMAs (opaque to hipcc's hazard recogniser) the result was wrong unless >= 32 cycles lay between.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_war.hip -o tools/ubench/mfma_war && tools/ubench/mfma_war
// Each variant: AHEAD independent MFMAs back to back (they occupy the pipe), then the MFMA under test, then NOPS wait states, then
// `v_mov` of a NaN into its SrcB (or SrcA) register.  The accumulator of the MFMA under test is compared with the first one's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NOP_0 ""
#define NOP_1 "s_nop 0\n"
#define NOP_2 "s_nop 1\n"
#define NOP_4 "s_nop 3\n"
#define NOP_8 "s_nop 7\n"
#define NOP_16 "s_nop 15\n"
#define NOP_24 "s_nop 15\ns_nop 7\n"
#define NOP_32 "s_nop 15\ns_nop 15\n"
#define NOP_48 "s_nop 15\ns_nop 15\ns_nop 15\n"
#define NOP_64 "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"

template <int V> __device__ void body(f32x16& r0, f32x16& r1, float a, float b, float a2, float b2);

#define VARIANT(ID, AHEAD_ASM, TEST_ASM, NOPS, KILL)                                                         \
    template <> __device__ void body<ID>(f32x16& r0, f32x16& r1, float a, float b, float a2, float b2) {      \
        f32x16 x0 = {}, x1 = {}, x2 = {}, t = {};                                                              \
        asm volatile(AHEAD_ASM TEST_ASM NOPS KILL "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"                  \
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(t), "+v"(a2), "+v"(b2) : "v"(a), "v"(b));            \
        r0 = x0; r1 = t;                                                                                       \
    }
#define AHEAD3 "v_mfma_f32_32x32x2_f32 %0, %6, %7, %0\nv_mfma_f32_32x32x2_f32 %1, %6, %7, %1\nv_mfma_f32_32x32x2_f32 %2, %6, %7, %2\n"
#define AHEAD1 "v_mfma_f32_32x32x2_f32 %0, %6, %7, %0\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"      /* the pipe is idle again */
#define TESTB "v_mfma_f32_32x32x2_f32 %3, %6, %5, %3\n"
#define TESTA "v_mfma_f32_32x32x2_f32 %3, %4, %7, %3\n"
#define KILLB "v_mov_b32 %5, 0x7fc00000\n"
#define KILLA "v_mov_b32 %4, 0x7fc00000\n"

VARIANT(0, AHEAD3, TESTB, NOP_0, KILLB)  VARIANT(1, AHEAD3, TESTB, NOP_1, KILLB)  VARIANT(2, AHEAD3, TESTB, NOP_2, KILLB)
VARIANT(3, AHEAD3, TESTB, NOP_4, KILLB)  VARIANT(4, AHEAD3, TESTB, NOP_8, KILLB)  VARIANT(5, AHEAD3, TESTB, NOP_16, KILLB)
VARIANT(6, AHEAD3, TESTB, NOP_24, KILLB) VARIANT(7, AHEAD3, TESTB, NOP_32, KILLB) VARIANT(8, AHEAD3, TESTB, NOP_48, KILLB)
VARIANT(9, AHEAD3, TESTB, NOP_64, KILLB)
VARIANT(10, AHEAD3, TESTA, NOP_0, KILLA) VARIANT(11, AHEAD3, TESTA, NOP_4, KILLA) VARIANT(12, AHEAD3, TESTA, NOP_16, KILLA)
VARIANT(13, AHEAD3, TESTA, NOP_32, KILLA) VARIANT(14, AHEAD3, TESTA, NOP_64, KILLA)
VARIANT(15, AHEAD1, TESTB, NOP_0, KILLB) VARIANT(16, AHEAD1, TESTB, NOP_4, KILLB) VARIANT(17, AHEAD1, TESTB, NOP_16, KILLB)
VARIANT(18, AHEAD1, TESTA, NOP_0, KILLA)
constexpr int NV = 19;

template <int V> __global__ void run(float* out) {
    const int lane = threadIdx.x;
    const float a = 1.f + 0.01f * lane, b = 2.f - 0.02f * lane;
    f32x16 r0, r1;
    body<V>(r0, r1, a, b, a, b);
    int bad = 0;
    for (int i = 0; i < 16; ++i) bad += !(r0[i] == r1[i]);
    out[V * 64 + lane] = (float)bad;
}

template <int V> void launch(float* d) { hipLaunchKernelGGL(run<V>, dim3(1), dim3(64), 0, 0, d); }
template <int... I> void launch_all(float* d, std::integer_sequence<int, I...>) { (launch<I>(d), ...); }

int main() {
    float* d; hipMalloc(&d, NV * 64 * 4);
    launch_all(d, std::make_integer_sequence<int, NV>{});
    float h[NV * 64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[NV] = {"B busy-pipe nops 0", "B busy 1", "B busy 2", "B busy 4", "B busy 8", "B busy 16", "B busy 24", "B busy 32", "B busy 48", "B busy 64",
                             "A busy 0", "A busy 4", "A busy 16", "A busy 32", "A busy 64", "B idle-pipe 0", "B idle 4", "B idle 16", "A idle 0"};
    for (int v = 0; v < NV; ++v) {
        int bad = 0; for (int l = 0; l < 64; ++l) bad += (int)h[v * 64 + l];
        printf("%-22s wrong accumulator values: %4d of 1024\n", names[v], bad);
    }
    return 0;
}
