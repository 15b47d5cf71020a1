// Where does the halo-tile weight-gradient kernel's time go?  Builds csrc/wgrad_halo.hip with -DWGH_ABL=<bits>.
//   bits: 1 = no next-tile global loads (address math + 15 loads per tile)   2 = no per-tile LDS restaging / barriers
#include <stdarg.h>
#include "../../self-supervised-anomaly-detection_amd/csrc/wgrad_halo.hip"
void ssad_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int ssad_bn_finalize_partials(const double*, int, int64_t, int, float, float, float*, float*, float*, float*, void*) { return 0; }
static void run(int64_t N, int H, int W, int Cin, int Cout) {
    const size_t nx = (size_t)N * H * W * Cin, ny = (size_t)N * H * W * Cout;
    const int splits = ssad_wgrad3x3_halo_splits(N, H, W, Cin, Cout);
    float *x, *dz, *slab;
    hipMalloc(&x, nx * 4); hipMalloc(&dz, ny * 4); hipMalloc(&slab, (size_t)splits * Cout * 9 * Cin * 4);
    hipMemset(x, 0, nx * 4); hipMemset(dz, 0, ny * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) ssad_conv_wgrad3x3_halo(dz, x, slab, splits, N, H, W, Cin, Cout, nullptr);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) ssad_conv_wgrad3x3_halo(dz, x, slab, splits, N, H, W, Cin, Cout, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("ABL %d  N=%lld %dx%d %d->%d splits %d: %.3f ms  %.1f TFLOP/s\n", WGH_ABL, (long long)N, H, W, Cin, Cout, splits, ms, 2.0 * N * H * W * Cin * Cout * 9 / ms / 1e9);
    hipFree(x); hipFree(dz); hipFree(slab);
}
int main() {
    run(256, 64, 64, 64, 64);
    run(256, 16, 16, 256, 256);
    return 0;
}
