// Developer probe: per-phase cycle stamps of conv3x3_halo_kernel on one dense-scatter step (48x48, step s).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCIAOSR_PROBE -I ciaosr_amd/csrc tools/conv_probe.hip -o gpurun_out/conv_probe
#include "../ciaosr_amd/csrc/runtime.hip"
#include "../ciaosr_amd/csrc/conv_f32.hip"
#include <vector>
#include <cstdio>
using namespace ciaosr;
int main(int argc, char** argv) {
    const int H = 48, W = 48, NL = 8, step = argc > 1 ? atoi(argv[1]) : 0;
    const int ldx = 576, M = H * W;
    float *X, *wgt, *bias, *acc, *part;
    hipMalloc(&X, (size_t)M * ldx * 4); hipMemset(X, 0, (size_t)M * ldx * 4);
    hipMalloc(&wgt, (size_t)512 * 576 * 4); hipMemset(wgt, 0, (size_t)512 * 576 * 4);
    hipMalloc(&bias, 512 * 4); hipMemset(bias, 0, 512 * 4);
    hipMalloc(&acc, (size_t)M * 512 * 4); hipMemset(acc, 0, (size_t)M * 512 * 4);
    hipMalloc(&part, (size_t)16 * M * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) dense_scatter_step(X, ldx, H, W, step, NL, wgt, bias, acc, part, (size_t)16 * M * 64, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int it = 0; it < 20; ++it) dense_scatter_step(X, ldx, H, W, step, NL, wgt, bias, acc, part, (size_t)16 * M * 64, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("step %d: %.2f us per launch (back-to-back)\n", step, ms * 1000 / 20);
    std::vector<unsigned long long> h(4096 * 8);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_probe), h.size() * 8);
    const int nwg = 72 * 2 * (NL - step) > 4096 ? 4096 : 72 * 2 * (NL - step);
    unsigned long long tmin = ~0ull, tmax = 0;
    double d[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nwg; ++b) {
        const unsigned long long* s = &h[b * 8];
        if (s[0] < tmin) tmin = s[0];
        if (s[6] > tmax) tmax = s[6];
        for (int k = 0; k < 6; ++k) d[k] += (double)(s[k + 1] - s[k]);
    }
    printf("WGs %d; kernel span (first start -> last end) %llu cycles\n", nwg, tmax - tmin);
    const char* nm[6] = {"B0 issue + halo load issue", "store B0 + wait + sync", "stage0 MFMA", "stages 1..2", "sync+kslice reduce", "epilogue"};
    for (int k = 0; k < 6; ++k) printf("  %-28s %8.0f cycles avg\n", nm[k], d[k] / nwg);
    // start-time distribution
    std::vector<unsigned long long> st(nwg);
    for (int b = 0; b < nwg; ++b) st[b] = h[b * 8] - tmin;
    unsigned long long late = 0; for (auto v : st) if (v > late) late = v;
    printf("  last WG starts %llu cycles after the first\n", late);
    return 0;
}
