// Calibration: the fp32 matrix-core issue rate this chip really sustains (v_mfma_f32_32x32x2_f32 from registers, no memory
// traffic), for 1 / 2 waves per SIMD and 2 / 4 independent accumulators, over a duration comparable to a training kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void spin(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 123.456f) out[0] = s;
}

template <int NACC>
static void run(int wgs_per_cu, int iters) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(spin<NACC>, dim3(grid), dim3(256), 0, 0, out, iters / 10);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 16 * NACC * (2.0 * 32 * 32 * 2);
        printf("acc %d  waves/SIMD %d  %.3f ms  %.1f TFLOP/s  (%.3f of 157.3; implied clock at full issue %.3f GHz)\n", NACC, wgs_per_cu, ms,
               flops / ms / 1e9, flops / ms / 1e9 / 157.3, flops / ms / 1e9 / 157.3 * 2.4);
    }
    hipFree(out);
}

int main() {
    run<2>(1, 4000);
    run<4>(1, 2000);
    run<2>(2, 4000);
    run<4>(2, 2000);
    run<4>(2, 20000);     // ~10 ms: long enough for the power controller to settle
    return 0;
}
