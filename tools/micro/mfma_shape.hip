// Does the per-piece staging cost of the fp32 implicit GEMM (profiles/r03_igemm_phases.md: ~90 matrix-pipe cycles per 16-byte-per-lane
// global load + LDS store, 8 pieces per 64 MFMAs) depend on the MFMA SHAPE?  v_mfma_f32_32x32x2_f32 moves 16 accumulator registers
// in and out per 64 cycles, v_mfma_f32_16x16x4_f32 four per 32 cycles: half the register-file traffic per FLOP.
//
// One workgroup of 256 threads per CU (LDS padding), the K-step of conv_igemm.hip's 128 x 128 x 32 tile in miniature:
//   SHAPE 0: 64 x v_mfma_f32_32x32x2_f32 per wave and K-step (4 accumulators of 16 registers)
//   SHAPE 1: 128 x v_mfma_f32_16x16x4_f32 (16 accumulators of 4 registers) -- the same FLOPs, the same 4 096 pipe cycles
// with PIECES 16-byte-per-lane global loads (L2-resident source) + ds_write_b128 of the piece loaded one K-step earlier spread between
// the MFMAs, FRAGS ds_read_b128 fragment reads per K-step, and one barrier.  Prints cycles per K-step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int PIECES, int FRAGS, int BARRIER>
__global__ __launch_bounds__(256, 1) void k(float* out, const f32x4* src, unsigned long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc32[4];
    f32x4 acc16[16];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc32[j][e] = 0.f;
    for (int j = 0; j < 16; ++j)
        for (int e = 0; e < 4; ++e) acc16[j][e] = 0.f;
    const f32x4* gp = src + tid + (size_t)(blockIdx.x & 255) * 256 * 64;       // 256 KB per workgroup, re-read: L2-resident
    f32x4 stage[PIECES > 0 ? PIECES : 1];
    for (int q = 0; q < PIECES; ++q) stage[q] = gp[q * 256];
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        f32x4 fa[4], fb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { fa[q] = *(const f32x4*)(lds + tid * 4 + q * 1024); fb[q] = *(const f32x4*)(lds + 8192 + tid * 4 + q * 1024); }
        constexpr int NM = SHAPE == 0 ? 64 : 128;                              // MFMAs per K-step
        constexpr int GAP = PIECES > 0 ? NM / (2 * PIECES) : NM;                // loads in the first half, stores in the second
        const f32x4* g2 = gp + (size_t)((it + 1) & 31) * 256 * 2;
#pragma clang loop unroll(full)
        for (int m = 0; m < NM; ++m) {
            if (SHAPE == 0) {
                acc32[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[(m >> 2) & 3][m & 3], fb[(m >> 4) & 3][m & 3], acc32[m & 3], 0, 0, 0);
            } else {
                acc16[m & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[(m >> 2) & 3][m & 3], fb[(m >> 5) & 3][m & 3], acc16[m & 15], 0, 0, 0);
            }
            if (PIECES > 0 && m < NM / 2 && m % GAP == GAP - 1) {
                __builtin_amdgcn_sched_barrier(0);
                const int q = m / GAP;
                *(f32x4*)(lds + ((it & 1) ? 4096 : 12288) + tid * 4 + (q & 3) * 1024 * 0) = stage[q];        // store last step's piece
                stage[q] = g2[q * 256];                                                                      // request the next one
                __builtin_amdgcn_sched_barrier(0);
            }
            if (FRAGS > 4 && m % (NM / 4) == NM / 4 - 1 && m + 1 < NM) {       // fragment reads of the next chunk
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 2; ++q) { fa[(m / (NM / 4) + q) & 3] = *(const f32x4*)(lds + tid * 4 + q * 1024 + 16); fb[(m / (NM / 4) + q) & 3] = *(const f32x4*)(lds + 8192 + tid * 4 + q * 1024 + 16); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (BARRIER) __syncthreads();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) s += acc32[j][e];
    for (int j = 0; j < 16; ++j)
        for (int e = 0; e < 4; ++e) s += acc16[j][e];
    for (int q = 0; q < PIECES; ++q) s += stage[q][0];
    if (s == 123.456f) out[0] = s;
    if (tid == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

template <int SHAPE, int PIECES, int FRAGS, int BARRIER>
static void run(const char* name, float* out, const f32x4* src, unsigned long long* clk) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k<SHAPE, PIECES, FRAGS, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, PIECES, FRAGS, BARRIER>), dim3(256), dim3(256), 100 * 1024, 0, out, src, clk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h;
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double fl = 256.0 * 4 * iters * 64 * (2.0 * 32 * 32 * 2);
    printf("%-58s %8.1f cycles per K-step (4096 = pipe)  %6.1f TFLOP/s\n", name, (double)h / iters, fl / ms / 1e9);
}

int main() {
    float* out;
    f32x4* src;
    unsigned long long* clk;
    hipMalloc(&out, 64); hipMalloc(&clk, 16);
    const size_t n = (size_t)256 * 256 * 64 + 64 * 256 * 2 + 4096;
    hipMalloc(&src, n * 16);
    float* h = (float*)malloc(n * 16);
    unsigned s = 12345;
    for (size_t i = 0; i < n * 4; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    hipMemcpy(src, h, n * 16, hipMemcpyHostToDevice);
    run<0, 0, 0, 0>("32x32x2  MFMAs only", out, src, clk);
    run<1, 0, 0, 0>("16x16x4  MFMAs only", out, src, clk);
    run<0, 0, 16, 1>("32x32x2  + fragment reads + barrier", out, src, clk);
    run<1, 0, 16, 1>("16x16x4  + fragment reads + barrier", out, src, clk);
    run<0, 8, 16, 1>("32x32x2  + 8 pieces + fragment reads + barrier", out, src, clk);
    run<1, 8, 16, 1>("16x16x4  + 8 pieces + fragment reads + barrier", out, src, clk);
    run<0, 16, 16, 1>("32x32x2  + 16 pieces + fragment reads + barrier", out, src, clk);
    run<1, 16, 16, 1>("16x16x4  + 16 pieces + fragment reads + barrier", out, src, clk);
    run<0, 8, 0, 0>("32x32x2  + 8 pieces only", out, src, clk);
    run<1, 8, 0, 0>("16x16x4  + 8 pieces only", out, src, clk);
    return 0;
}
