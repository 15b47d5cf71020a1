// What costs clock?  fp32 MFMA stream (2 waves per SIMD, 4 accumulators) with LDS / global traffic mixed in at the rates of
// the conv kernels; reports TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime x 100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only.  1: one ds_read_b32 per MFMA feeding its operand.  2: one ds_read_b128 per 4 MFMAs.  3: as 1 with random
// (non-zero) LDS contents.  4: as 0 with random register operands.  5: MODE 1 + one global_load_dwordx4 per 16 MFMAs.
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* src, unsigned long long* clk, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (MODE == 3 || MODE == 5 || MODE == 6) ? src[i] : 0.f;
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = MODE == 4 ? src[threadIdx.x] : 0.f, b = (MODE == 4 || MODE == 3 || MODE == 5 || MODE == 6) ? src[threadIdx.x + 256] : 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    const float* lp = lds + threadIdx.x;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    const f32x4* gp = (const f32x4*)src + threadIdx.x + (size_t)blockIdx.x * 256;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 2) {
                const f32x4 v = *(const f32x4*)(lds + threadIdx.x * 4 + u * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j] + a, b, acc[j], 0, 0, 0);
            } else if (MODE == 6) {           // both operands from LDS, random data: 2 ds_read_b32 per MFMA
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = lp[(u * 4 + j) * 256], w = lp[4096 + (u * 4 + j) * 256];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, w, acc[j], 0, 0, 0);
                }
            } else if (MODE == 1 || MODE == 3 || MODE == 5) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = lp[(u * 4 + j) * 256];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, b, acc[j], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            }
        }
        if (MODE == 5) { g += gp[(i & 1023) * 256 * 0 + ((i & 255) * 65536)]; }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = g[0] + g[1];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int MODE>
static void run(const char* name, float* out, const float* src, unsigned long long* clk) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, out, src, clk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double fl = 512.0 * 4 * iters * 16 * (2.0 * 32 * 32 * 2);
    printf("%-64s %7.3f ms  %6.1f TFLOP/s  clock %.3f GHz\n", name, ms, fl / ms / 1e9, (double)h[0] / (double)h[1] * 0.1);
}

int main() {
    float *out, *src;
    unsigned long long* clk;
    hipMalloc(&out, 64); hipMalloc(&clk, 16);
    const size_t n = (size_t)256 * 65536 * 4 + 512 * 256 * 4 + 8192;      // floats reachable by MODE 5
    hipMalloc(&src, n * 4);
    float* h = (float*)malloc(n * 4);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    run<0>("MFMA only, zero operands", out, src, clk);
    run<4>("MFMA only, random operands", out, src, clk);
    run<1>("MFMA + ds_read_b32 per MFMA, zeros", out, src, clk);
    run<3>("MFMA + ds_read_b32 per MFMA, random data", out, src, clk);
    run<2>("MFMA + ds_read_b128 per 4 MFMAs (+4 v_add), zeros", out, src, clk);
    run<6>("MFMA + 2 ds_read_b32 per MFMA (both operands), random data", out, src, clk);
    run<5>("MFMA + ds_read_b32 per MFMA + global_load_dwordx4 per 16, random", out, src, clk);
    return 0;
}
