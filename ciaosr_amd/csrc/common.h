// Shared host-side helpers for libciaosr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <mutex>

#include "../../include/ciaosr_hip.h"

namespace ciaosr {

constexpr int kWave = 64;  // CDNA wavefront

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- opt-in per-kernel event timing (bench.py roofline leg) ------------------------------------
// When enabled via ciaosr_prof_enable(1), every launch wrapper brackets its kernel with two HIP
// events recorded on the launch stream.  ciaosr_prof_collect() synchronises and folds the elapsed
// times into per-kernel totals.  Off by default: zero overhead besides one branch.
struct ProfScope {
    int slot;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    ProfScope(const char* name, hipStream_t s);
    ~ProfScope();
};

// > 64 KiB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel.  The attribute belongs to the
// (kernel, device) pair: `granted` (zero-initialised, owned by the call site, one per kernel) holds the largest size already
// granted on each device, and the attribute is raised again whenever a launch needs more (a kernel whose LDS size depends on
// its arguments, e.g. the weights-resident 1x1 with K = the block width).  Raising is serialised so that the attribute never
// shrinks under a concurrent caller.  Returns false if HIP refuses.
constexpr int kMaxDevices = 32;
struct LdsAttrOnce {
    std::atomic<unsigned> granted[kMaxDevices];
    std::mutex raise;
};
template <typename K>
static inline bool allow_big_lds(LdsAttrOnce& once, K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return false;
    if (once.granted[dev].load(std::memory_order_acquire) >= bytes) return true;
    std::lock_guard<std::mutex> lock(once.raise);
    if (once.granted[dev].load(std::memory_order_relaxed) >= bytes) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) !=
        hipSuccess)
        return false;
    once.granted[dev].store((unsigned)bytes, std::memory_order_release);
    return true;
}

// Ordering point of an exchange through LDS between the lanes of ONE wave (accumulator layout -> row-major transposes in the store
// epilogues): the hardware runs a wave's LDS instructions in order, so no instruction is needed -- the wavefront-scope fences and the
// wave barrier only forbid the COMPILER to move the read phase above the write phase (or a rewrite of the buffer above its last read).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

int launch_status(const char* what);

// ciaosr_options_t hygiene, checked by every entry point that takes one: unknown route bits and out-of-range fields are refused
// (a caller built against a later layout -- or passing garbage -- is refused instead of being half-understood).
static inline bool options_ok(const ciaosr_options_t* o) {
    if (!o) return true;
    return o->query_grid_w >= 0 && (o->csa_attn_tile128 == 0 || o->csa_attn_tile128 == 1) && (o->head_route & ~127) == 0 && (o->kv_rows == 0 || o->kv_rows == 32 || o->kv_rows == 64) &&
           (o->decode_rows == 0 || o->decode_rows == 32 || o->decode_rows == 64) && (o->bf16_single == 0 || o->bf16_single == 1) &&
           (o->dense_direct >= 0 && o->dense_direct <= 2) && (o->csa_scores_gemm == 0 || o->csa_scores_gemm == 1) &&
           o->f16_pairs >= 0 && o->f16_pairs <= 3;
}

}  // namespace ciaosr

// one-time (per device) > 64 KiB LDS opt-in for `kernel`; returns CIAOSR_ERR_LAUNCH from the enclosing function on failure
#define CIAOSR_BIG_LDS(kernel, bytes)                                                              \
    do {                                                                                           \
        static ::ciaosr::LdsAttrOnce once_;                                                        \
        if (!::ciaosr::allow_big_lds(once_, kernel, bytes)) return CIAOSR_ERR_LAUNCH;              \
    } while (0)

#define CIAOSR_CHECK_ARG(cond)                                                         \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            std::fprintf(stderr, "[ciaosr_hip] bad argument: %s (%s:%d)\n", #cond,     \
                         __FILE__, __LINE__);                                          \
            return CIAOSR_ERR_BAD_ARG;                                                 \
        }                                                                              \
    } while (0)
