// Developer probe: per-phase cycle stamps of dense_scatter_small_kernel on one dense-scatter step (48x48, step s).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCIAOSR_PROBE -I ciaosr_amd/csrc -I include tools/scatter_probe.hip -o gpurun_out/scatter_probe
#include "../ciaosr_amd/csrc/runtime.hip"
#include "../ciaosr_amd/csrc/dense_scatter_f32.hip"
#include <vector>
#include <algorithm>
#include <utility>
#include <cstdio>
using namespace ciaosr;
int main(int argc, char** argv) {
    const int H = 48, W = 48, NL = 8, step = argc > 1 ? atoi(argv[1]) : 0;
    const int ldx = 576, M = H * W;
    float *X, *frag, *bias, *acc;
    hipMalloc(&X, (size_t)M * ldx * 4); hipMemset(X, 0, (size_t)M * ldx * 4);
    hipMalloc(&frag, (size_t)512 * 576 * 4); hipMemset(frag, 0, (size_t)512 * 576 * 4);
    hipMalloc(&bias, 512 * 4); hipMemset(bias, 0, 512 * 4);
    hipMalloc(&acc, (size_t)M * 512 * 4); hipMemset(acc, 0, (size_t)M * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) dense_scatter_small(X, ldx, H, W, step, NL, frag, bias, acc, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int it = 0; it < 1; ++it) dense_scatter_small(X, ldx, H, W, step, NL, frag, bias, acc, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("step %d: %.2f us per launch (back-to-back)\n", step, ms * 1000 / 1);
    std::vector<unsigned long long> h(4096 * 8);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_sprobe), h.size() * 8);
    const int nwg = 36 * 2 * (NL - step);
    unsigned long long tmin = ~0ull, tmax = 0;
    double d[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < nwg; ++b) {
        const unsigned long long* s = &h[b * 8];
        if (s[0] < tmin) tmin = s[0];
        if (s[5] > tmax) tmax = s[5];
        for (int k = 0; k < 5; ++k) d[k] += (double)(s[k + 1] - s[k]);
    }
    printf("WGs %d; kernel span (first WG start -> last WG end) %llu ticks\n", nwg, tmax - tmin);
    const char* nm[5] = {"issue weights + prev", "issue patch, store, sync", "MFMA loop", "sync + reduce write", "reduce read + epilogue"};
    for (int k = 0; k < 5; ++k) printf("  %-28s %8.0f ticks avg\n", nm[k], d[k] / nwg);
    // per-CU timelines: HW_ID bits: [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se (gfx9 layout)
    {
        std::vector<std::pair<unsigned long long, int>> order;
        for (int b = 0; b < nwg; ++b) {
            const unsigned hw = (unsigned)h[b * 8 + 6], xcc = (unsigned)h[b * 8 + 7] & 0xF;
            const unsigned long long key = ((unsigned long long)xcc << 32) | (hw & 0xFF00u);
            order.push_back({key, b});
        }
        std::sort(order.begin(), order.end());
        int shown = 0;
        for (size_t i = 0; i < order.size() && shown < 12; ) {
            size_t j = i;
            while (j < order.size() && order[j].first == order[i].first) ++j;
            if (j - i >= 2) {
                printf("  CU key %llx: %zu WGs:", order[i].first, j - i);
                for (size_t k = i; k < j; ++k) {
                    const unsigned long long* s = &h[order[k].second * 8];
                    printf("  [wg %d: start %llu mfma %llu..%llu end %llu]", order[k].second, s[0] - tmin, s[2] - tmin, s[3] - tmin, s[5] - tmin);
                }
                printf("\n");
                ++shown;
            }
            i = j;
        }
    }
    unsigned long long late = 0;
    for (int b = 0; b < nwg; ++b) if (h[b * 8] - tmin > late) late = h[b * 8] - tmin;
    printf("  last WG starts %llu ticks after the first\n", late);
    return 0;
}
