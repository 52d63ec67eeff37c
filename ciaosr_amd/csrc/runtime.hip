// Library runtime: version, error strings, launch status, opt-in per-kernel event timing.
#include <cstring>
#include "common.h"

#include <mutex>
#include <string>
#include <vector>

namespace ciaosr {

namespace {
struct Pending {
    int slot;
    hipEvent_t a, b;
};
std::mutex g_mu;
bool g_enabled = false;
std::vector<std::string> g_names;
std::vector<double> g_total_ms;
std::vector<long> g_count;
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_pool;
std::string g_filter;  // empty = every kernel

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

ProfScope::ProfScope(const char* name, hipStream_t s) : slot(-1), stream(s), ev0(nullptr), ev1(nullptr) {
    if (!g_enabled) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_filter.empty() && g_filter != name) return;
    for (size_t i = 0; i < g_names.size(); ++i)
        if (g_names[i] == name) { slot = (int)i; break; }
    if (slot < 0) {
        slot = (int)g_names.size();
        g_names.emplace_back(name);
        g_total_ms.push_back(0.0);
        g_count.push_back(0);
    }
    ev0 = get_event();
    ev1 = get_event();
    (void)hipEventRecord(ev0, stream);
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    (void)hipEventRecord(ev1, stream);
    std::lock_guard<std::mutex> lk(g_mu);
    g_pending.push_back({slot, ev0, ev1});
}

int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::fprintf(stderr, "[ciaosr_hip] launch of %s failed: %s\n", what, hipGetErrorString(e));
        return CIAOSR_ERR_LAUNCH;
    }
    return CIAOSR_OK;
}

}  // namespace ciaosr

using namespace ciaosr;

extern "C" int ciaosr_version(void) { return 210; }

// sizeof of every struct of the ABI, by name: lets a binding (ciaosr_amd/_lib.py's ctypes mirrors) verify its layout
extern "C" size_t ciaosr_sizeof(const char* type_name) {
    if (!type_name) return 0;
    const struct { const char* n; size_t s; } tab[] = {
        {"ciaosr_options_t", sizeof(ciaosr_options_t)},
        {"ciaosr_csattn_weights_t", sizeof(ciaosr_csattn_weights_t)},
        {"ciaosr_mlp_t", sizeof(ciaosr_mlp_t)},
        {"ciaosr_head_weights_t", sizeof(ciaosr_head_weights_t)},
        {"ciaosr_conv_t", sizeof(ciaosr_conv_t)},
        {"ciaosr_rdn_weights_t", sizeof(ciaosr_rdn_weights_t)},
        {"ciaosr_edsr_weights_t", sizeof(ciaosr_edsr_weights_t)},
        {"ciaosr_swin_block_t", sizeof(ciaosr_swin_block_t)},
        {"ciaosr_swinir_weights_t", sizeof(ciaosr_swinir_weights_t)},
    };
    for (const auto& e : tab)
        if (std::strcmp(e.n, type_name) == 0) return e.s;
    return 0;
}

extern "C" const char* ciaosr_error_string(int code) {
    switch (code) {
        case CIAOSR_OK: return "ok";
        case CIAOSR_ERR_BAD_ARG: return "bad argument (dims / alignment / null pointer)";
        case CIAOSR_ERR_LAUNCH: return "HIP kernel launch failed";
        case CIAOSR_ERR_UNSUPPORTED: return "unsupported configuration";
        case CIAOSR_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

extern "C" int ciaosr_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_enabled = on != 0;
    return CIAOSR_OK;
}

extern "C" int ciaosr_prof_filter(const char* kernel) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_filter = kernel ? kernel : "";
    return CIAOSR_OK;
}

extern "C" int ciaosr_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& p : g_pending) { g_pool.push_back(p.a); g_pool.push_back(p.b); }
    g_pending.clear();
    for (auto& v : g_total_ms) v = 0.0;
    for (auto& v : g_count) v = 0;
    return CIAOSR_OK;
}

extern "C" int ciaosr_prof_collect(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& p : g_pending) {
        (void)hipEventSynchronize(p.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_total_ms[p.slot] += ms;
            g_count[p.slot] += 1;
        }
        g_pool.push_back(p.a);
        g_pool.push_back(p.b);
    }
    g_pending.clear();
    return CIAOSR_OK;
}

extern "C" int ciaosr_prof_get(const char* kernel, double* total_ms, long* launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_names.size(); ++i)
        if (g_names[i] == kernel) {
            if (total_ms) *total_ms = g_total_ms[i];
            if (launches) *launches = g_count[i];
            return CIAOSR_OK;
        }
    if (total_ms) *total_ms = 0.0;
    if (launches) *launches = 0;
    return CIAOSR_ERR_BAD_ARG;
}

extern "C" int ciaosr_prof_names(char* buf, int buflen) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::string s;
    for (auto& n : g_names) { s += n; s += ';'; }
    if ((int)s.size() + 1 > buflen) return CIAOSR_ERR_BAD_ARG;
    std::snprintf(buf, buflen, "%s", s.c_str());
    return CIAOSR_OK;
}
