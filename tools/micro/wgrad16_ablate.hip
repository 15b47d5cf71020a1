// Where does the half-tensor weight-gradient kernel (csrc/wgrad16.hip) spend its time?  Built with -DWG16_ABL=<bits> (see the kernel
// file) and timed on the trunk shapes at batch 256:  bash tools/micro/wgrad16_ablate.sh > gpurun_out/wgrad16_ablate.log
#include <stdarg.h>
#include "../../self-supervised-anomaly-detection_amd/csrc/wgrad16.hip"

void ssad_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

static void run(int64_t N, int H, int W, int Cin, int Cout, int S) {
    const int Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
    const size_t ex = (size_t)N * H * W * Cin, ez = (size_t)N * Ho * Wo * Cout;
    hf *x, *dz; float* slab;
    const int splits = ssad_wgrad3x3_g16_splits(N, Ho, Wo, Cin, Cout, S);
    hipMalloc(&x, ex * 2); hipMalloc(&dz, ez * 2); hipMalloc(&slab, (size_t)splits * Cout * 9 * Cin * 4);
    hipMemset(x, 0, ex * 2); hipMemset(dz, 0, ez * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() { return ssad_conv_wgrad3x3_g16_h(dz, x, slab, splits, N, Ho, Wo, H, W, Cin, Cout, S, (int64_t)ez, nullptr); };
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
    const double fl = 2.0 * N * Ho * Wo * Cout * 9 * Cin;
    printf("ABL %2d  N=%lld %dx%d %d->%d s%d (%d splits): %.1f us  %.0f TFLOP/s\n", WG16_ABL, (long long)N, H, W, Cin, Cout, S, splits, ms * 1e3,
           fl / ms / 1e9);
    hipFree(x); hipFree(dz); hipFree(slab);
}

int main() {
    run(256, 64, 64, 64, 64, 1);
    run(256, 32, 32, 128, 128, 1);
    run(256, 16, 16, 256, 256, 1);
    run(256, 8, 8, 512, 512, 1);
    run(256, 64, 64, 64, 128, 2);
    return 0;
}
