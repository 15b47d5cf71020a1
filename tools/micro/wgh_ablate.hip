// Where does the halo-tile weight-gradient kernel's time go?  Builds csrc/wgrad_halo.hip with -DWGH_ABL=<bits>.
//   bits: 1 = no next-tile global loads   2 = no per-tile LDS restaging / barriers   4 = no LDS fragment reads in the matrix loop
#include <stdarg.h>
#include <stdlib.h>
#include <vector>
#include "../../self-supervised-anomaly-detection_amd/csrc/wgrad_halo.hip"
void ssad_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int ssad_bn_finalize_partials(const double*, int, int64_t, int, float, float, float*, float*, float*, float*, void*) { return 0; }
static void run(int64_t N, int H, int W, int Cin, int Cout) {
    const size_t nx = (size_t)N * H * W * Cin, ny = (size_t)N * H * W * Cout;
    const int splits = ssad_wgrad3x3_halo_splits(N, H, W, Cin, Cout);
    float *x, *dz, *slab;
    hipMalloc(&x, nx * 4); hipMalloc(&dz, ny * 4); hipMalloc(&slab, (size_t)splits * Cout * 9 * Cin * 4);
    hipMemset(x, 0, nx * 4); hipMemset(dz, 0, ny * 4);
    if (getenv("WGH_RANDOM")) {          // non-zero operands (the matrix cores draw more power on real data)
        std::vector<float> hx(nx), hy(ny);
        unsigned sd = 1;
        for (auto& v : hx) { sd = sd * 1664525u + 1013904223u; v = ((sd >> 8) & 0xffff) / 65536.0f - 0.5f; }
        for (auto& v : hy) { sd = sd * 1664525u + 1013904223u; v = ((sd >> 8) & 0xffff) / 65536.0f - 0.5f; }
        hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(dz, hy.data(), ny * 4, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) ssad_conv_wgrad3x3_halo(dz, x, slab, splits, N, H, W, Cin, Cout, (int64_t)N * H * W * Cout, nullptr);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) ssad_conv_wgrad3x3_halo(dz, x, slab, splits, N, H, W, Cin, Cout, (int64_t)N * H * W * Cout, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
#if WGH_ABL & 16
    {
        const int nwg = (Cin / 64) * (Cout / 64) * splits;
        unsigned long long* tr;
        hipMalloc(&tr, (size_t)nwg * 72 * 8);
        hipMemset(tr, 0, (size_t)nwg * 72 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_wgh_trace), &tr, sizeof(tr));
        ssad_conv_wgrad3x3_halo(dz, x, slab, splits, N, H, W, Cin, Cout, (int64_t)N * H * W * Cout, nullptr);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)nwg * 72);
        hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long* none = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_wgh_trace), &none, sizeof(none));
        hipFree(tr);
        {   // spread of the workgroups in real time (100 MHz ticks)
            unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
            for (int b = 0; b < nwg; ++b) {
                const unsigned long long a = h[(size_t)b * 72 + 64], z = h[(size_t)b * 72 + 65];
                s0 = a < s0 ? a : s0; s1 = a > s1 ? a : s1; e0 = z < e0 ? z : e0; e1 = z > e1 ? z : e1;
            }
            printf("  %d workgroups: starts spread over %.1f us, ends over %.1f us, first start -> last end %.1f us\n", nwg, (s1 - s0) * 0.01, (e1 - e0) * 0.01,
                   (e1 - s0) * 0.01);
        }
        {   // duration by XCD and by shader engine
            double sum[8] = {0}, sq[8] = {0}; int cnt[8] = {0};
            double mn[8], mx[8];
            for (int i = 0; i < 8; ++i) { mn[i] = 1e30; mx[i] = 0; }
            for (int b = 0; b < nwg; ++b) {
                const unsigned long long* t = &h[(size_t)b * 72];
                const int xcc = (int)((t[66] >> 32) & 7);
                const double us = (t[65] - t[64]) * 0.01;
                sum[xcc] += us; cnt[xcc]++; mn[xcc] = us < mn[xcc] ? us : mn[xcc]; mx[xcc] = us > mx[xcc] ? us : mx[xcc];
                sq[xcc] += (double)(t[63] - t[0]) / (us * 1e3);
            }
            for (int i = 0; i < 8; ++i)
                if (cnt[i]) printf("    XCD %d: %3d workgroups, duration mean %.1f us (min %.1f, max %.1f), clock %.3f GHz\n", i, cnt[i], sum[i] / cnt[i], mn[i], mx[i], sq[i] / cnt[i]);
        }
        for (int b = 0; b < 2; ++b) {
            const unsigned long long* t = &h[(size_t)b * 72];
            printf("  wg %d: start->first tile %llu; tiles (restage, mfma):", b, t[1] - t[0]);
            for (int i = 0; i < 30 && t[2 + 2 * i]; ++i) {
                const unsigned long long nxt = t[3 + 2 * i] ? t[3 + 2 * i] : t[62];
                printf(" (%llu, %llu)", t[2 + 2 * i] - t[1 + 2 * i], nxt - t[2 + 2 * i]);
            }
            printf("; epilogue %llu; total %llu cycles in %.1f us = %.3f GHz\n", t[63] - t[62], t[63] - t[0], (t[65] - t[64]) * 0.01,
                   (double)(t[63] - t[0]) / (double)(t[65] - t[64]) * 0.1);
        }
    }
#endif
    printf("ABL %d  N=%lld %dx%d %d->%d splits %d: %.3f ms  %.1f TFLOP/s\n", WGH_ABL, (long long)N, H, W, Cin, Cout, splits, ms, 2.0 * N * H * W * Cin * Cout * 9 / ms / 1e9);
    hipFree(x); hipFree(dz); hipFree(slab);
}
int main() {
    run(256, 64, 64, 64, 64);
    run(256, 16, 16, 256, 256);
    return 0;
}
