// K-loop schedule study of the implicit-GEMM conv: builds csrc/conv_igemm.hip with -DIGEMM_VAR=<v> and times forward (NHWC and
// position-major) and dgrad shapes of the two phases on uniform random operands.
//   for v in 0 1 3; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iself-supervised-anomaly-detection_amd/csrc \
//       -DIGEMM_VAR=$v tools/micro/igemm_var.hip -o tools/micro/build/ig_var$v; done      (cross-compiles without a GPU)
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../../self-supervised-anomaly-detection_amd/csrc/conv_igemm.hip"
void ssad_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int ssad_bn_finalize_partials(const double*, int, int64_t, int, float, float, float*, float*, float*, float*, void*) { return 0; }
bool ssad_linear_small_ok(const void*, const void*, int64_t, int) { return false; }
int ssad_linear_small_launch(const float*, const float*, float*, const float*, const float*, const float*, int, int, int, int, double*, int*, void*, int) { return 1; }
static double inb(int h, int k, int s, int p) {
    int ho = (h + 2 * p - k) / s + 1, n = 0;
    for (int o = 0; o < ho; ++o) for (int t = 0; t < k; ++t) n += (unsigned)(o * s - p + t) < (unsigned)h;
    return n;
}
static float* dev_rand(size_t n, unsigned seed, bool relu) {
    std::vector<float> h(n);
    unsigned sd = seed;
    for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = ((sd >> 8) & 0xffff) / 65536.0f - 0.5f; if (relu && v < 0) v = 0; }
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
}
static void run(const char* name, int64_t N, int H, int W, int Cin, int Cout, int k, int s, int p, int mode) {   // mode 0 nhwc, 1 hwnc, 2 dgrad
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    const size_t nx = (size_t)N * H * W * Cin, ny = (size_t)N * Ho * Wo * Cout, nw = (size_t)Cout * k * k * Cin;
    float* x = dev_rand(nx, 1, true); float* w = dev_rand(nw, 7, false); float* y; float* dyv = nullptr;
    hipMalloc(&y, (mode == 2 ? nx : ny) * 4);
    if (mode == 2) dyv = dev_rand(ny, 3, false);
    float* aff = dev_rand(2 * (size_t)Cout, 11, false);       // mode 3: a ring launch of the shared layer1 (folded BatchNorm, residual, ReLU)
    float* resid = mode == 3 ? dev_rand(ny, 5, false) : nullptr;
    auto go = [&]() {
        if (mode == 3) return ssad_conv_igemm_fwd_hwnc_ring(x, w, y, aff, aff + Cout, resid, 1, N, H, W, Cin, Cout, k, k, s, p, 4, 12, nullptr);
        if (mode == 2) return ssad_conv_igemm_dgrad(dyv, w, y, nullptr, N, Ho, Wo, Cout, H, W, Cin, k, k, s, p, nullptr);
        return mode ? ssad_conv_igemm_fwd_hwnc(x, w, y, nullptr, nullptr, nullptr, 1, N, H, W, Cin, Cout, k, k, s, p, nullptr)
                    : ssad_conv_igemm_fwd(x, w, y, nullptr, nullptr, nullptr, 1, N, H, W, Cin, Cout, k, k, s, p, nullptr); };
    for (int i = 0; i < 3; ++i) if (go()) { printf("%s: launch refused\n", name); return; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int it = 20;
    hipEventRecord(e0);
    for (int i = 0; i < it; ++i) go();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    const double alg = 2.0 * N * Ho * Wo * Cin * Cout * k * k;
    const double exe = mode == 3 ? alg * (256 - 81) / 256.0 * 0.9 : mode == 1 ? 2.0 * N * Cin * Cout * inb(H, k, s, p) * inb(W, k, s, p) : alg;
#if IGEMM_TRACE
    {   // phase time line of the workgroups that ran on the CU of workgroup 0 (one extra launch)
        const size_t nwg = 1 << 20;
        unsigned long long* tr;
        hipMalloc(&tr, nwg * 64); hipMemset(tr, 0, nwg * 64);
        hipMemcpyToSymbol(HIP_SYMBOL(g_ig_trace), &tr, sizeof(tr));
        go(); hipDeviceSynchronize();
        std::vector<unsigned long long> hh(nwg * 8);
        hipMemcpy(hh.data(), tr, nwg * 64, hipMemcpyDeviceToHost);
        const unsigned long long key0 = hh[7] & 0xffffffff0000ff00ull;
        std::vector<size_t> mine;
        for (size_t b = 0; b < nwg; ++b) if (hh[b * 8] && (hh[b * 8 + 7] & 0xffffffff0000ff00ull) == key0) mine.push_back(b);
        std::sort(mine.begin(), mine.end(), [&](size_t a, size_t b) { return hh[a * 8] < hh[b * 8]; });
        const unsigned long long base = hh[mine[0] * 8];
        unsigned long long lo = ~0ull, hi = 0; size_t tot = 0;
        for (size_t b = 0; b < nwg; ++b) if (hh[b * 8]) { ++tot; }
        printf("  %s: %zu workgroups stamped, %zu on the CU of workgroup 0; cycles from its first start\n", name, tot, mine.size());
        for (size_t i = 0; i < mine.size() && i < 20; ++i) {
            const size_t b = mine[i];
            printf("  wg %7zu slot w%llu  start %8llu  prologue %6llu  kloop %7llu  epilogue %6llu  end %8llu\n", b, hh[b * 8 + 7] & 15, hh[b * 8] - base,
                   hh[b * 8 + 1] - hh[b * 8], hh[b * 8 + 2] - hh[b * 8 + 1], hh[b * 8 + 3] - hh[b * 8 + 2], hh[b * 8 + 3] - base);
        }
        (void)lo; (void)hi;
        unsigned long long* none = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_ig_trace), &none, sizeof(none));
        hipFree(tr);
    }
#endif
    // checksum so that variants can be compared bit for bit
    std::vector<float> out((mode == 2 ? nx : ny));
    hipMemcpy(out.data(), y, out.size() * 4, hipMemcpyDeviceToHost);
    unsigned cs = 0; for (float v : out) { unsigned u; memcpy(&u, &v, 4); cs = cs * 31u + u; }
    printf("VAR %d  %-34s %8.3f ms  alg %6.1f  exec %6.1f TFLOP/s  cs %08x\n", IGEMM_VAR, name, ms, alg / ms / 1e9, exe / ms / 1e9, cs);
    fflush(stdout);
    hipFree(x); hipFree(y); hipFree(w); if (dyv) hipFree(dyv);
}
int main() {
    if (getenv("L2_ONLY")) { run("score l2 107648x8x8 128>128", 107648, 8, 8, 128, 128, 3, 1, 1, 1); run("train l2 256x32x32 128>128", 256, 32, 32, 128, 128, 3, 1, 1, 0); return 0; }
    if (getenv("L34_ONLY")) { run("score l3 107648x4x4 256>256", 107648, 4, 4, 256, 256, 3, 1, 1, 1); run("score l4 107648x2x2 512>512", 107648, 2, 2, 512, 512, 3, 1, 1, 1); return 0; }
    if (getenv("RING_ONLY")) { run("score l1 ring 26912x16x16 64>64", 26912, 16, 16, 64, 64, 3, 1, 1, 3); return 0; }
    run("train l2 256x32x32 128>128", 256, 32, 32, 128, 128, 3, 1, 1, 0);
    run("train l3 256x16x16 256>256", 256, 16, 16, 256, 256, 3, 1, 1, 0);
    run("train l4 256x8x8 512>512", 256, 8, 8, 512, 512, 3, 1, 1, 0);
    run("train l2 dgrad 128>128", 256, 32, 32, 128, 128, 3, 1, 1, 2);
    run("train l3 s2 128>256 fwd", 256, 32, 32, 128, 256, 3, 2, 1, 0);
    run("train l3 s2 128>256 dgrad", 256, 32, 32, 128, 256, 3, 2, 1, 2);
    run("train l2 ds 1x1s2 64>128", 256, 64, 64, 64, 128, 1, 2, 0, 0);
    run("b32 l4 32x8x8 512>512", 32, 8, 8, 512, 512, 3, 1, 1, 0);
    run("b32 l3 32x16x16 256>256", 32, 16, 16, 256, 256, 3, 1, 1, 0);
    run("score l2 s2 16x16 64>128", 15979, 16, 16, 64, 128, 3, 2, 1, 1);
    run("score l2 15979x8x8 128>128", 15979, 8, 8, 128, 128, 3, 1, 1, 1);
    run("score l3 15979x4x4 256>256", 15979, 4, 4, 256, 256, 3, 1, 1, 1);
    run("score l4 15979x2x2 512>512", 15979, 2, 2, 512, 512, 3, 1, 1, 1);
    run("score l2 107648x8x8 128>128", 107648, 8, 8, 128, 128, 3, 1, 1, 1);
    run("score l1 ring 26912x16x16 64>64", 26912, 16, 16, 64, 64, 3, 1, 1, 3);
    return 0;
}
