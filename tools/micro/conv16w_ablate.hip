// Where does the register-fed half-tensor conv (csrc/conv16w.hip) spend its time?  Built with -DCONV16W_ABL=<bits> (see the kernel file) and timed
// on the four trunk shapes at batch 256:  bash tools/micro/conv16w_ablate.sh > gpurun_out/conv16w_ablate.log
#include <stdarg.h>
#include "../../self-supervised-anomaly-detection_amd/csrc/conv16w.hip"

void ssad_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int ssad_bn_finalize_partials(const double*, int, int64_t, int, float, float, float*, float*, float*, float*, void*) { return 0; }

static void run(int64_t N, int H, int W, int C) {
    const size_t elems = (size_t)N * H * W * C;
    hf *x, *y, *w;
    hipMalloc(&x, elems * 2); hipMalloc(&y, elems * 2); hipMalloc(&w, (size_t)C * 9 * C * 2);
    hipMemset(x, 0, elems * 2); hipMemset(w, 0, (size_t)C * 9 * C * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() {
        return ssad_conv3x3_hw(x, w, y, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, C, C, nullptr, 0.f, 0.f, nullptr, nullptr,
                              nullptr, nullptr, nullptr);
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
    const double fl = 2.0 * N * H * W * C * 9 * C;
    printf("ABL %2d  N=%lld %dx%dx%d: %.1f us  %.0f TFLOP/s  %.2f TB/s (in + out)\n", CONV16W_ABL, (long long)N, H, W, C, ms * 1e3, fl / ms / 1e9,
           4.0 * elems / ms / 1e9);
    hipFree(x); hipFree(y); hipFree(w);
}

int main() {
    run(256, 64, 64, 64);
    run(256, 32, 32, 128);
    run(256, 16, 16, 256);
    run(256, 8, 8, 512);
    return 0;
}
