// Shared host-side helpers for libciaosr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

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

// > 64 KiB of dynamic LDS needs the attribute once per kernel.
template <typename K>
static inline void allow_big_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int launch_status(const char* what);

}  // namespace ciaosr

#define CIAOSR_CHECK_ARG(cond)                                                         \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            std::fprintf(stderr, "[ciaosr_hip] bad argument: %s (%s:%d)\n", #cond,     \
                         __FILE__, __LINE__);                                          \
            return CIAOSR_ERR_BAD_ARG;                                                 \
        }                                                                              \
    } while (0)
