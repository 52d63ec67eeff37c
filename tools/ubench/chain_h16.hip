// Feasibility probe for the weights-stationary, register-chained 16-bit MLP chain: ONE persistent workgroup of 4 waves per
// CU (one wave per SIMD); a 64-KB slot of pre-packed f16 weight fragments is DMA'd (buffer_load ... lds) into one of two LDS buffers while
// the waves run the other; a wave owns M row tiles of 32 rows for ALL 256 columns of a 256 x 256 layer, the accumulators of layer l
// become, after relu + cvt_pk, the B-operand fragments of layer l + 1 (swapped operands: weights = A) -- activations never touch LDS.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench/chain_h16.hip -o tools/ubench/chain_h16 && tools/ubench/chain_h16
// Prints cycles per layer and wave against the MFMA issue time (8 n-tiles x 16 k-steps x M x 32 cycles), with and without the DMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int SLOT = 64 * 1024;
typedef const __attribute__((address_space(3))) unsigned char* lds_cptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4* lds_c128;

__device__ __forceinline__ unsigned pack_relu2(float a, float b) {
    const f32x2 v = {__builtin_amdgcn_fmed3f(a, 0.f, 65504.f), __builtin_amdgcn_fmed3f(b, 0.f, 65504.f)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// one 64-KB slot = 4 n-tiles x 16 k-steps of 1-KB fragments; out tiles T0 .. T0 + 3
template <int M, typename DMAF>
__device__ __forceinline__ void half_layer(lds_cptr slot, int lane, int T0, const u32x4 (&in)[M][16], u32x4 (&out)[M][16],
                                           const __attribute__((address_space(3))) float* bias, DMAF dma_piece) {
    lds_cptr fr = slot + lane * 16;
    f32x16 acc[2][M];
    f32x16 bv[2];
    u32x4 a[3];
    auto load_bias = [&](int T, f32x16& b) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 v = *(const __attribute__((address_space(3))) f32x4*)(bias + 32 * T + 8 * g + 4 * (lane >> 5));
            b[4 * g] = v.x; b[4 * g + 1] = v.y; b[4 * g + 2] = v.z; b[4 * g + 3] = v.w;
        }
    };
    load_bias(T0, bv[0]);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int cur = nt & 1;
        lds_cptr ft = fr + nt * 16384;          // one base register per n-tile, immediate offsets inside it
        asm volatile("" : "+v"(ft));
        a[0] = *(lds_c128)(ft);
        a[1] = *(lds_c128)(ft + 1024);
        if (nt + 1 < 4) load_bias(T0 + nt + 1, bv[cur ^ 1]);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (ks + 2 < 16) a[(ks + 2) % 3] = *(lds_c128)(ft + (ks + 2) * 1024);
#pragma unroll
            for (int mi = 0; mi < M; ++mi) acc[cur][mi] = mfma(a[ks % 3], in[mi][ks], ks == 0 ? bv[cur] : acc[cur][mi]);
            // epilogue of the previous n-tile: 4 accumulator registers (two packed registers) per k-step from k-step 2 on
            if (nt > 0 && ks >= 2 && ks < 2 + 4 * M) {
                const int piece = ks - 2, mi = piece >> 2, qd = piece & 3;
                const f32x16& c = acc[cur ^ 1][mi];
                unsigned o0 = pack_relu2(c[4 * qd + 0], c[4 * qd + 1]), o1 = pack_relu2(c[4 * qd + 2], c[4 * qd + 3]);
                asm volatile("" : "+v"(o0), "+v"(o1));      // computed HERE (hipcc otherwise sinks every epilogue behind the half layer)
                u32x4& o = out[mi][2 * (T0 + nt - 1) + (qd >> 1)];
                if (qd & 1) { o.z = o0; o.w = o1; } else { o.x = o0; o.y = o1; }
            }
            if (ks >= 10 && ks < 14) dma_piece(4 * nt + ks - 10);      // this wave's 16 pieces of the slot after the next: one behind each of 16 k-steps
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int mi = 0; mi < M; ++mi)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f32x16& c = acc[1][mi];
            u32x4 o;
            o.x = pack_relu2(c[8 * hf + 0], c[8 * hf + 1]); o.y = pack_relu2(c[8 * hf + 2], c[8 * hf + 3]);
            o.z = pack_relu2(c[8 * hf + 4], c[8 * hf + 5]); o.w = pack_relu2(c[8 * hf + 6], c[8 * hf + 7]);
            out[mi][2 * (T0 + 3) + hf] = o;
        }
}

template <int M, bool DMA>
__global__ __launch_bounds__(256, 1) void chain(const unsigned char* __restrict__ Wg, unsigned w_bytes, int n_slots, const float* __restrict__ bias,
                                                 int passes, unsigned* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __attribute__((address_space(3))) float* lbias = (__attribute__((address_space(3))) float*)((lds_cptr)lds + 2 * SLOT);
    for (int i = threadIdx.x; i < 512; i += 256) lbias[i] = bias[i];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const i32x4 desc = {(int)(unsigned)(size_t)Wg, (int)(((size_t)Wg >> 32) & 0xFFFFu), (int)w_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // four 1-KB pieces per M0 setting: the instruction offset applies to the source AND the LDS destination
    auto dma4 = [&](unsigned lds_dst, unsigned voff, unsigned soff) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds\n\tbuffer_load_dwordx4 %0, %2, %3 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %0, %2, %3 offen offset:2048 lds\n\tbuffer_load_dwordx4 %0, %2, %3 offen offset:3072 lds"
                     :: "v"(voff), "s"(lds_dst), "s"(desc), "s"(soff) : "memory");
    };
    // wave w brings pieces 16 w .. 16 w + 15 of a slot
    auto issue_slot = [&](int slot_idx, int buf) {
        const unsigned voff = (unsigned)(16 * w) * 1024u + (unsigned)lane * 16u;
        const unsigned soff = (unsigned)(slot_idx % n_slots) * (unsigned)SLOT;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf * SLOT + (unsigned)(16 * w) * 1024u);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma4(dst + i * 4096u, voff, soff + i * 4096u);
    };
    const unsigned voff_w = (unsigned)(16 * w) * 1024u + (unsigned)lane * 16u;
    auto dma1 = [&](unsigned lds_dst, unsigned soff) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" :: "v"(voff_w), "s"(lds_dst), "s"(desc), "s"(soff) : "memory");
    };
    u32x4 A[M][16], B[M][16];
#pragma unroll
    for (int mi = 0; mi < M; ++mi)
#pragma unroll
        for (int k = 0; k < 16; ++k) A[mi][k] = u32x4{0x3c003c00u + lane + mi, 0x38003800u + k, 0x34003400u, 0x30003000u};
    const int total = passes * 2 * 2;           // passes x 2 layers x 2 halves
    if (DMA) { issue_slot(0, 0); issue_slot(1, 1); }
    else { issue_slot(0, 0); issue_slot(1, 1); }
    unsigned long long t0 = 0;
    int sidx = 0;
#pragma unroll 1
    for (int p = 0; p < passes; ++p) {
        if (p == 1) t0 = __builtin_readcyclecounter();
        // layer X: A -> B, layer Y: B -> A; each two halves
#pragma unroll
        for (int ly = 0; ly < 2; ++ly) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int buf = hf;            // slot index parity == half
                if (DMA) {
                    if (sidx + 1 < total) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if (sidx < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                // the slot after the next goes into the buffer the PREVIOUS slot used (every wave is past it: barrier above), piece by piece
                // behind this half layer's MFMAs
                const bool go = DMA && sidx >= 1 && sidx + 1 < total;
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf ^ 1) * SLOT + (unsigned)(16 * w) * 1024u);
                const unsigned soff = (unsigned)((sidx + 1) % n_slots) * (unsigned)SLOT;
                auto piece = [&](int i) { if (go) dma1(dst + i * 1024u, soff + i * 1024u); };
                if (ly == 0) half_layer<M>((lds_cptr)lds + buf * SLOT, lane, 4 * hf, A, B, lbias, piece);
                else half_layer<M>((lds_cptr)lds + buf * SLOT, lane, 4 * hf, B, A, lbias + 256, piece);
                ++sidx;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
#pragma unroll
    for (int mi = 0; mi < M; ++mi)
#pragma unroll
        for (int k = 0; k < 16; ++k) s += A[mi][k].x ^ A[mi][k].y ^ A[mi][k].z ^ A[mi][k].w;
    out[blockIdx.x * 256 + t] = s;
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int M, bool DMA>
static void run(const unsigned char* W, unsigned wb, int n_slots, const float* bias, unsigned* out, unsigned long long* cyc, int passes) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain<M, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT + 2048);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((chain<M, DMA>), dim3(256), dim3(256), 2 * SLOT + 2048, 0, W, wb, n_slots, bias, passes, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
    const double layers = 2.0 * (passes - 1), mfma_cyc = 8.0 * 16 * M * 32;
    const double flop = 256.0 * 4 * passes * 2 * (2.0 * 256 * 256 * 32 * M);
    printf("M=%d dma=%d: %.3f ms, %.0f cycles per layer and wave (MFMA issue %.0f: %.3f), %.1f TFLOP/s, err=%s\n", M, (int)DMA, best, avg / layers, mfma_cyc,
           mfma_cyc / (avg / layers), flop / best / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int n_slots = 17;                  // one pass of the real head: 6 hidden layers + the 640-column output layer = 17 slots
    unsigned char* W; float* bias; unsigned* out; unsigned long long* cyc;
    hipMalloc(&W, (size_t)n_slots * SLOT); hipMalloc(&bias, 4096); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    std::vector<unsigned short> hw((size_t)n_slots * SLOT / 2);
    for (auto& v : hw) v = (unsigned short)(0x2000 + (rand() & 0x3ff) + ((rand() & 1) << 15));     // ~ +-0.01
    hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 4096);
    const int passes = 40;
    run<2, false>(W, n_slots * SLOT, n_slots, bias, out, cyc, passes);
    run<2, true>(W, n_slots * SLOT, n_slots, bias, out, cyc, passes);
    if (getenv("CHAIN_M3")) {   // 3 row tiles per wave: hipcc spills 170-230 registers (384 activation + 96 accumulator registers of 512)
        run<3, false>(W, n_slots * SLOT, n_slots, bias, out, cyc, passes);
        run<3, true>(W, n_slots * SLOT, n_slots, bias, out, cyc, passes);
    }
    return 0;
}
