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
                             const float* residual, int relu, int M, int K, int N, double* stats, int* stat_rows, void* stream,
                             int round = 0);      // round: operands rounded to bf16 (1) / fp16 (2) as they are loaded

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Storage type of an activation tensor: float, or _Float16 for the precision-16 step whose tensors live in HBM as halves (what
// torch.autocast stores under the reference's pl.Trainer(precision=16), tools.py:263).  Arithmetic is fp32 either way: four
// consecutive elements are read as / written from an f32x4 (16-byte or 8-byte access).
typedef _Float16 hf;
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 ld4<hf>(const hf* p) {
    const f16x4 v = *(const f16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void st4<hf>(hf* p, f32x4 v) {
    *(f16x4*)p = f16x4{(hf)v[0], (hf)v[1], (hf)v[2], (hf)v[3]};
}
// One lane's share of a streaming (HBM-bound) kernel is ONE 16-byte access per tensor: four floats or eight halves (round 5: the half
// instantiations first read 8 bytes per lane -- twice the instructions per byte -- and ran BELOW the fp32 kernels' byte rate).
typedef float f32x8 __attribute__((ext_vector_type(8)));
template <typename T> struct Lane { static constexpr int E = 4; using vec = f32x4; };
template <> struct Lane<hf> { static constexpr int E = 8; using vec = f32x8; };
template <typename T> __device__ __forceinline__ typename Lane<T>::vec ldv(const T* p);
template <> __device__ __forceinline__ f32x4 ldv<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x8 ldv<hf>(const hf* p) { return __builtin_convertvector(*(const f16x8*)p, f32x8); }
template <typename T> __device__ __forceinline__ void stv(T* p, typename Lane<T>::vec v);
template <> __device__ __forceinline__ void stv<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void stv<hf>(hf* p, f32x8 v) { *(f16x8*)p = __builtin_convertvector(v, f16x8); }
// The same 16 bytes WITHOUT the conversion: a kernel that requests several tensors under run-time conditions loads raw pieces first and
// converts afterwards -- with ldv inside an `if (ptr)` hipcc puts the wait for that load (its convert) into the branch and the
// requests of one iteration go out one after the other (round 6: the half column reductions ran at 3.0-3.8 TB/s for that reason).
template <typename T> struct RawLane { using t = f32x4; };
template <> struct RawLane<hf> { using t = f16x8; };
template <typename T> __device__ __forceinline__ typename RawLane<T>::t ldrawv(const T* p) { return *(const typename RawLane<T>::t*)p; }
__device__ __forceinline__ f32x4 cvtraw(f32x4 v) { return v; }
__device__ __forceinline__ f32x8 cvtraw(f16x8 v) { return __builtin_convertvector(v, f32x8); }
// E consecutive per-channel parameters (fp32 either way), E = 4 or 8, index in units of E
template <typename V> __device__ __forceinline__ V ldpar(const float* p, int i);
template <> __device__ __forceinline__ f32x4 ldpar<f32x4>(const float* p, int i) { return ((const f32x4*)p)[i]; }
template <> __device__ __forceinline__ f32x8 ldpar<f32x8>(const float* p, int i) {
    const f32x4 a = ((const f32x4*)p)[2 * i], b = ((const f32x4*)p)[2 * i + 1];
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// four consecutive elements in their storage type (no conversion): what a kernel stages when its operands are the stored halves
template <typename T> struct Raw4 { using t = f32x4; };
template <> struct Raw4<hf> { using t = f16x4; };
template <typename T> __device__ __forceinline__ typename Raw4<T>::t ldraw4(const T* p) { return *(const typename Raw4<T>::t*)p; }
// the value a tensor of type T holds after storing v (statistics of a half tensor are those of the stored halves)
template <typename T> __device__ __forceinline__ float stored(float v) { return v; }
template <> __device__ __forceinline__ float stored<hf>(float v) { return (float)(hf)v; }

// v_mfma_f32_32x32x2_f32: lane l feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l&31].  Exact f32 fma chain in k order.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
