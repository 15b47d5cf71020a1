// What does the implicit-GEMM conv pay for its epilogue?  Builds csrc/conv_igemm.hip with -DIGEMM_ABL=<bits> (1 = no epilogue).
//   for a in 0 1; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iself-supervised-anomaly-detection_amd/csrc -DIGEMM_ABL=$a \
//       tools/micro/igemm_ablate.hip -o /tmp/ig_$a && /tmp/ig_$a; done
#include <stdarg.h>
#include <stdlib.h>
#include <vector>
#include "../../self-supervised-anomaly-detection_amd/csrc/conv_igemm.hip"
void ssad_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int ssad_bn_finalize_partials(const double*, int, int64_t, int, float, float, float*, float*, float*, float*, void*) { return 0; }
static void run(const char* name, int64_t N, int H, int W, int Cin, int Cout, int hwnc) {
    const size_t nx = (size_t)N * H * W * Cin, ny = (size_t)N * H * W * Cout, nw = (size_t)Cout * 9 * Cin;
    float *x, *y, *w;
    hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&w, nw * 4);
    std::vector<float> h(nx > nw ? nx : nw);
    unsigned sd = 1;
    for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = ((sd >> 8) & 0xffff) / 65536.0f - 0.5f; }
    hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, h.data(), nw * 4, hipMemcpyHostToDevice);
    auto go = [&]() { return hwnc ? ssad_conv_igemm_fwd_hwnc(x, w, y, nullptr, nullptr, nullptr, 1, N, H, W, Cin, Cout, 3, 3, 1, 1, nullptr)
                                  : ssad_conv_igemm_fwd(x, w, y, nullptr, nullptr, nullptr, 1, N, H, W, Cin, Cout, 3, 3, 1, 1, nullptr); };
    for (int i = 0; i < 3; ++i) go();
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) go();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("ABL %d  %-28s %.3f ms  %.1f TFLOP/s (algorithmic)\n", IGEMM_ABL, name, ms, 2.0 * N * H * W * Cin * Cout * 9 / ms / 1e9);
    hipFree(x); hipFree(y); hipFree(w);
}
int main() {
    run("train l2 256x32x32 128>128", 256, 32, 32, 128, 128, 0);
    run("train l4 256x8x8 512>512", 256, 8, 8, 512, 512, 0);
    run("score l2 15979x8x8 128>128", 15979, 8, 8, 128, 128, 1);
    run("score l3 15979x4x4 256>256", 15979, 4, 4, 256, 256, 1);
    return 0;
}
