// Shared helpers for the gfx950 kernels (wave64, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

void ssad_set_error(const char* fmt, ...);

#define SSAD_CHECK_ARG(cond, msg)                                     \
    do {                                                              \
        if (!(cond)) {                                                \
            ssad_set_error("%s: %s (%s)", __func__, msg, #cond);      \
            return 2;                                                 \
        }                                                             \
    } while (0)

#define SSAD_CHECK_LAUNCH()                                           \
    do {                                                              \
        hipError_t e_ = hipGetLastError();                            \
        if (e_ != hipSuccess) {                                       \
            ssad_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return 1;                                                 \
        }                                                             \
    } while (0)

// Opt a kernel in to more than 64 KB of dynamic LDS; a refusal (a device that is not gfx950: 160 KB of LDS per CU) is an error
// the caller sees, not a launch that fails later.
#define SSAD_SET_DYN_LDS(kernel, bytes)                                                                      \
    do {                                                                                                     \
        hipError_t e_ = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
        if (e_ != hipSuccess) {                                                                              \
            ssad_set_error("%s: %d bytes of LDS per workgroup refused (%s); this library targets gfx950",    \
                           __func__, (int)(bytes), hipGetErrorString(e_));                                   \
            return 1;                                                                                        \
        }                                                                                                    \
    } while (0)

// train.hip: mean / invstd / running statistics from `nblk` rows of [2][C] double partial sums over R samples
int ssad_bn_finalize_partials(const double* partial, int nblk, int64_t R, int C, float eps, float momentum, float* mean,
                              float* invstd, float* running_mean, float* running_var, void* stream);

// linear_small.hip: linear layers over few rows (training batches of the projection head); conv_igemm.hip's entry points route
// 1 x 1 layers on 1 x 1 maps there when ssad_linear_small_ok says so
bool ssad_linear_small_ok(const void* a, const void* b, int64_t M, int K);
int ssad_linear_small_launch(const float* a, const float* b, float* y, const float* scale, const float* shift,
                             const float* residual, int relu, int M, int K, int N, double* stats, int* stat_rows, void* stream);

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// v_mfma_f32_32x32x2_f32: lane l feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l&31].  Exact f32 fma chain in k order.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
