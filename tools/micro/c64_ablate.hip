// Where does the halo-tile conv kernel's time go?  Builds csrc/conv_c64.hip with -DC64_ABL=<bits> and times it.
//   bits: 1 = no per-tap weight restaging (no global loads, LDS stores, barriers in the tap loop)
//         2 = no LDS fragment reads in the tap loop      4 = no epilogue      8 = no halo loads from HBM
//   for a in 0 1 2 3 4 8 15; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iself-supervised-anomaly-detection_amd/csrc \
//       -DC64_ABL=$a tools/micro/c64_ablate.hip -o /tmp/c64_abl_$a && /tmp/c64_abl_$a; done
#include <stdarg.h>
#include <vector>
#include <algorithm>
#include "../../self-supervised-anomaly-detection_amd/csrc/conv_c64.hip"

void ssad_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int ssad_bn_finalize_partials(const double*, int, int64_t, int, float, float, float*, float*, float*, float*, void*) { return 0; }

static void run(int64_t N, int H, int W, int eval) {
    const size_t elems = (size_t)N * H * W * 64;
    float *x, *y, *w, *sc;
    hipMalloc(&x, elems * 4); hipMalloc(&y, elems * 4); hipMalloc(&w, 64 * 9 * 64 * 4); hipMalloc(&sc, 64 * 4);
    hipMemset(x, 0, elems * 4); hipMemset(w, 0, 64 * 9 * 64 * 4); hipMemset(sc, 0, 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() {
        if (eval) return ssad_conv3x3_c64_eval(x, w, y, sc, sc, nullptr, 1, N, H, W, 0, 0, 0, nullptr);
        return ssad_conv3x3_c64(x, w, y, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, nullptr, 0.f, 0.f,
                                nullptr, nullptr, nullptr, nullptr, nullptr);
    };
    for (int i = 0; i < 3; ++i) go();
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) go();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
#if C64_ABL & 16
    {   // phase time line of the workgroups that ran on one CU (last timed launch)
        const int64_t nwg = N * ((H + 7) / 8) * ((W + 15) / 16);
        unsigned long long* tr;
        hipMalloc(&tr, nwg * 64);
        hipMemset(tr, 0, nwg * 64);
        hipMemcpyToSymbol(HIP_SYMBOL(g_c64_trace), &tr, sizeof(tr));
        go();
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(nwg * 8);
        hipMemcpy(h.data(), tr, nwg * 64, hipMemcpyDeviceToHost);
        const unsigned long long key0 = h[7] & 0xffffffff0000ff00ull;
        std::vector<int64_t> mine;
        for (int64_t b = 0; b < nwg; ++b)
            if ((h[b * 8 + 7] & 0xffffffff0000ff00ull) == key0) mine.push_back(b);
        std::sort(mine.begin(), mine.end(), [&](int64_t a, int64_t b) { return h[a * 8] < h[b * 8]; });
        const unsigned long long base = h[mine[0] * 8];
        printf("  CU of workgroup 0: %zu workgroups; columns: wg slot start fill_done taps_done stored (cycles from the first start)\n", mine.size());
        for (size_t i = 0; i < mine.size() && i < 24; ++i) {
            const int64_t b = mine[i];
            printf("  %7lld w%llu %9llu %9llu %9llu %9llu   fill %6llu (issue %5llu landed+lds %6llu barrier %6llu) taps %6llu epi %6llu\n", (long long)b, h[b * 8 + 7] & 15, h[b * 8] - base, h[b * 8 + 1] - base,
                   h[b * 8 + 2] - base, h[b * 8 + 3] - base, h[b * 8 + 1] - h[b * 8], h[b * 8 + 4] - h[b * 8], h[b * 8 + 5] - h[b * 8 + 4], h[b * 8 + 1] - h[b * 8 + 5],
                   h[b * 8 + 2] - h[b * 8 + 1], h[b * 8 + 3] - h[b * 8 + 2]);
        }
        unsigned long long* none = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_c64_trace), &none, sizeof(none));
        hipFree(tr);
    }
#endif
    const double fl = 2.0 * N * H * W * 64 * 9 * 64;
    printf("ABL %2d  %s N=%lld %dx%d: %.3f ms  %.1f TFLOP/s\n", C64_ABL, eval ? "eval " : "train", (long long)N, H, W, ms, fl / ms / 1e9);
    hipFree(x); hipFree(y); hipFree(w); hipFree(sc);
}

int main() {
    run(256, 64, 64, 0);
    run(15979, 16, 16, 1);
    return 0;
}
