// Time stamps (s_memrealtime, 10 ns) inside one workgroup of the register-fed conv (csrc/conv16w.hip built with -DCONV16W_TRACE=<workgroup>):
// every wave of it; matrix waves -- 0/1 around the first barrier, 2/3 around each fill's barrier, 4/5 around a tile's epilogue; stager waves -- 101 start of
// a fill's work, 102 after the LDS writes, 103 after the load requests (then the barrier).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iself-supervised-anomaly-detection_amd/csrc -DCONV16W_TRACE=17 tools/micro/conv32w_trace.hip -o /tmp/conv32w_trace
#include <stdarg.h>
#include <vector>
#include "../../self-supervised-anomaly-detection_amd/csrc/conv16w.hip"

void ssad_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int ssad_bn_finalize_partials(const double*, int, int64_t, int, float, float, float*, float*, float*, float*, void*) { return 0; }

template <typename T>
static void run(int64_t N, int H, int W, int C, bool random) {
    const size_t elems = (size_t)N * H * W * C;
    T *x, *y, *w;
    hipMalloc(&x, elems * sizeof(T)); hipMalloc(&y, elems * sizeof(T)); hipMalloc(&w, (size_t)C * 9 * C * sizeof(T));
    if (random) {
        std::vector<T> hx(elems);
        unsigned s = 12345u;
        for (size_t i = 0; i < elems; ++i) { s = s * 1664525u + 1013904223u; hx[i] = (T)(((int)(s >> 8) % 2001 - 1000) * 1e-3f); }
        hipMemcpy(x, hx.data(), elems * sizeof(T), hipMemcpyHostToDevice);
        hipMemcpy(w, hx.data(), (size_t)C * 9 * C * sizeof(T), hipMemcpyHostToDevice);
    } else {
        hipMemset(x, 0, elems * sizeof(T)); hipMemset(w, 0, (size_t)C * 9 * C * sizeof(T));
    }
    long long* tr;
    hipMalloc(&tr, 8192 * 2 * sizeof(long long));
    conv16w_trace_buf = tr;
    auto go = [&]() {
        if constexpr (std::is_same<T, float>::value)
            return ssad_conv3x3_fw(x, w, y, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, C, C, nullptr, 0.f, 0.f, nullptr,
                                   nullptr, nullptr, nullptr, nullptr);
        else
            return ssad_conv3x3_hw(x, w, y, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, C, C, nullptr, 0.f, 0.f, nullptr,
                                   nullptr, nullptr, nullptr, nullptr);
    };
    for (int i = 0; i < 3; ++i) go();
    hipDeviceSynchronize();
    hipMemset(tr, 0, 8192 * 2 * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    go();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(8192 * 2);
    hipMemcpy(h.data(), tr, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    printf("== %s N=%lld %dx%dx%d %s: launch %.1f us\n", sizeof(T) == 4 ? "float" : "half", (long long)N, H, W, C, random ? "random" : "zeros", ms * 1e3);
    // per wave: time of every stamp relative to the workgroup's first one
    long long t0 = 0;
    for (int w = 0; w < 8; ++w) if (h[w * 2048 + 1] && (!t0 || h[w * 2048 + 1] < t0)) t0 = h[w * 2048 + 1];
    for (int w = 0; w < 4; ++w) {
        const long long* g = h.data() + w * 2048;
        double work = 0, wait = 0, epi = 0, first = 0;
        int nf = 0, nt = 0;
        long long last = g[1], tend = g[1], lastc = g[0] >> 8;
        double wcyc = 0;
        std::vector<double> works, waits, epis, arrive;
        for (int i = 0; i < 1020 && (i == 0 || g[2 * i + 1]); ++i) {
            const long long tag = g[2 * i] & 255, t = g[2 * i + 1], cyc = g[2 * i] >> 8;
            const double d = (t - last) * 0.01;
            if (tag == 1) first = d;
            if (tag == 2) { work += d; works.push_back(d); ++nf; arrive.push_back((t - t0) * 0.01); wcyc += (double)(cyc - lastc); }
            lastc = cyc;
            if (tag == 3) { wait += d; waits.push_back(d); }
            if (tag == 5) { epi += d; epis.push_back(d); ++nt; }
            if (tag == 4) work += d;
            last = t; tend = t;
        }
        printf("matrix wave %d: end %.1f us; first barrier %.2f; %d fills: work %.1f (%.2f each), barrier waits %.1f (%.2f each); %d epilogues %.1f (%.2f each)\n",
               w, (tend - t0) * 0.01, first, nf, work, work / (nf ? nf : 1), wait, wait / (nf ? nf : 1), nt, epi, epi / (nt ? nt : 1));
        printf("  cycles per fill of work %.0f (s_memtime) = %.3f GHz\n", wcyc / (nf ? nf : 1), wcyc / (work > 0 ? work : 1) * 1e-3);
        printf("  work:"); for (size_t i = 0; i < works.size() && i < 12; ++i) printf(" %.2f", works[i]); printf("\n");
        printf("  wait:"); for (size_t i = 0; i < waits.size() && i < 12; ++i) printf(" %.2f", waits[i]); printf("\n");
        printf("  arrive:"); for (size_t i = 0; i < arrive.size() && i < 12; ++i) printf(" %.2f", arrive[i]); printf("\n");
        printf("  epilogues:"); for (size_t i = 0; i < epis.size() && i < 12; ++i) printf(" %.2f", epis[i]); printf("\n");
    }
    for (int w = 4; w < 8; w += 3) {
        const long long* g = h.data() + w * 2048;
        double wr = 0, ld = 0, wt = 0; int ns = 0;
        long long last = g[1];
        std::vector<double> wrs, lds_, arrive;
        for (int i = 0; i < 1020 && (i == 0 || g[2 * i + 1]); ++i) {
            const long long tag = g[2 * i] & 255, t = g[2 * i + 1];
            const double d = (t - last) * 0.01;
            if (tag == 101 && i) wt += d;
            if (tag == 102) { wr += d; wrs.push_back(d); ++ns; }
            if (tag == 103) { ld += d; lds_.push_back(d); arrive.push_back((t - t0) * 0.01); }
            last = t;
        }
        printf("stager wave %d: %d fills: LDS writes %.1f (%.2f each), load requests %.1f (%.2f each), barrier waits %.1f\n", w, ns, wr, wr / (ns ? ns : 1), ld,
               ld / (ns ? ns : 1), wt);
        printf("  writes:"); for (size_t i = 0; i < wrs.size() && i < 12; ++i) printf(" %.2f", wrs[i]); printf("\n");
        printf("  loads:"); for (size_t i = 0; i < lds_.size() && i < 12; ++i) printf(" %.2f", lds_[i]); printf("\n");
        printf("  arrive:"); for (size_t i = 0; i < arrive.size() && i < 12; ++i) printf(" %.2f", arrive[i]); printf("\n");
    }
    hipFree(x); hipFree(y); hipFree(w); hipFree(tr);
}

int main() {
    run<float>(256, 64, 64, 64, false);
    run<float>(256, 32, 32, 128, false);
    run<hf>(256, 64, 64, 64, true);
    return 0;
}
