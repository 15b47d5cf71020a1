// Does a wave streaming fp32 MFMAs starve the other wave on its SIMD?  512-thread workgroups (one per CU): waves 0-3 (one per
// SIMD) stream v_mfma_f32_32x32x2_f32, waves 4-7 run a fixed non-matrix workload and time it.
//   mode 0: partner idle (the MFMA half exits at once)   mode 1: partner streams MFMAs (2 accumulators)   mode 2: 4 accumulators
//   work 0: 2000 dependent v_add   1: 2000 independent v_add (4 chains)   2: 500 x (ds_write_b128 + ds_read_b128)
//   work 3: 200 dependent global loads (pointer chase in L2)   4: 64 x 16 independent global_load_dwordx4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__device__ void stream_mfma(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 123.456f) out[0] = s;
}

__global__ __launch_bounds__(512, 1) void k(float* out, const int* chase, const float* big, unsigned long long* t, int mode, int work, int prio, int iters) {
    __shared__ f32x4 lds[512];
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (mode == 1) stream_mfma<2>(out, iters);
        if (mode == 2) stream_mfma<4>(out, iters);
        return;
    }
    if (prio) __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_s_sleep(100);                 // let the partner get going
    const unsigned long long t0 = __builtin_readcyclecounter();
    float r = 0.f;
    if (work == 0) {
        float x = threadIdx.x;
        for (int i = 0; i < 2000; ++i) x = x + 1.25f;
        r = x;
    } else if (work == 1) {
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
        for (int i = 0; i < 500; ++i) { x0 += 1.25f; x1 += 1.5f; x2 += 1.75f; x3 += 2.f; }
        r = x0 + x1 + x2 + x3;
    } else if (work == 2) {
        f32x4 v = {1.f, 2.f, 3.f, 4.f};
        for (int i = 0; i < 500; ++i) { lds[threadIdx.x] = v; v = lds[(threadIdx.x + 64) & 511 | 256]; v[0] += 1.f; }
        r = v[0] + v[1];
    } else if (work == 3) {
        int p = threadIdx.x & 63;
        for (int i = 0; i < 200; ++i) p = chase[p];
        r = p;
    } else {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        const f32x4* src = (const f32x4*)big + (size_t)blockIdx.x * 65536 + (threadIdx.x & 255);
        for (int i = 0; i < 64; ++i) {
            f32x4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = src[((i * 16 + j) & 255) * 256];     // 256 x 256 f32x4 = the 1 MiB slice of this workgroup
#pragma unroll
            for (int j = 0; j < 16; ++j) s += v[j];
        }
        r = s[0] + s[1] + s[2] + s[3];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + wave - 4] = t1 - t0;
    if (r == 123.456f) out[1] = r;
}

int main() {
    float *out, *big;
    int* chase;
    unsigned long long* t;
    hipMalloc(&out, 64); hipMalloc(&chase, 64 * 4); hipMalloc(&t, 256 * 4 * 8);
    hipMalloc(&big, (size_t)256 * 65536 * 16 + 65536);
    hipMemset(big, 0, (size_t)256 * 65536 * 16);
    int h[64];
    for (int i = 0; i < 64; ++i) h[i] = (i * 17 + 5) & 63;
    hipMemcpy(chase, h, sizeof(h), hipMemcpyHostToDevice);
    const char* wn[] = {"2000 dependent v_add", "2000 v_add in 4 chains", "500 x ds_write_b128+ds_read_b128", "200 dependent L2 loads", "64 x 16 global_load_dwordx4"};
    for (int work = 0; work < 5; ++work)
        for (int mode = 0; mode < 3; ++mode)
            for (int prio = 0; prio < 2; ++prio) {
                if (mode == 0 && prio) continue;
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, chase, big, t, mode, work, prio, 4000);
                hipDeviceSynchronize();
                unsigned long long ht[1024];
                hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost);
                double s = 0;
                for (int i = 0; i < 1024; ++i) s += ht[i];
                printf("%-34s partner %-18s prio %d: %9.0f cycles\n", wn[work], mode == 0 ? "idle" : mode == 1 ? "MFMA x2 acc" : "MFMA x4 acc", prio, s / 1024);
            }
    return 0;
}
