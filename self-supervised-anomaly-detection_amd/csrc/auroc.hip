// Area under the ROC curve on the GPU: radix sort (hipCUB) + tie-aware rank sum.
//
// Replaces sklearn.metrics.roc_curve + auc as the reference calls them for pixel / image AUROC
// (src/self_supervised/metrics.py:49-56, src/self_supervised/tools.py:76-98) when the scores already live on the GPU
// (83 x 65 536 pixel scores for one MVTec category).  The trapezoidal ROC area equals the Mann-Whitney statistic
// with ties counted one half:
//     AUC = sum over positives i of ( #negatives with a smaller score + 0.5 * #negatives with an equal score ) / (P * N)
// computed exactly in integers/halves and reduced in fp64 in a fixed order (deterministic).
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace {

// after sorting by score ascending: start[i] = index of the first element of i's tie group, cneg = inclusive scan of negatives
__global__ void mark_kernel(const float* __restrict__ keys, const uint8_t* __restrict__ lab, int64_t n, int64_t* __restrict__ start,
                            int64_t* __restrict__ neg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    start[i] = (i == 0 || keys[i] != keys[i - 1]) ? i : 0;     // max-scan turns this into the group start
    neg[i] = lab[i] ? 0 : 1;
}

__global__ void end_kernel(const float* __restrict__ keys, int64_t n, int64_t* __restrict__ endm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // i enumerates the REVERSED array
    if (i >= n) return;
    const int64_t j = n - 1 - i;
    endm[i] = (j == n - 1 || keys[j] != keys[j + 1]) ? -j : -(n + 1);     // min over the suffix as max of negatives
}

// per-block partial sums of 2*(neg_below + 0.5 neg_equal) over positives (an integer), and positive counts
__global__ void contrib_kernel(const uint8_t* __restrict__ lab, const int64_t* __restrict__ start, const int64_t* __restrict__ endr,
                               const int64_t* __restrict__ cneg, int64_t n, double* __restrict__ partial) {
    __shared__ double s2[256];
    __shared__ double sp[256];
    double acc = 0, pos = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!lab[i]) continue;
        const int64_t gs = start[i], ge = -endr[n - 1 - i];
        const int64_t below = gs > 0 ? cneg[gs - 1] : 0;
        const int64_t equal = cneg[ge] - below;
        acc += (double)(2 * below + equal);
        pos += 1.0;
    }
    s2[threadIdx.x] = acc;
    sp[threadIdx.x] = pos;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s2[threadIdx.x] += s2[threadIdx.x + o]; sp[threadIdx.x] += sp[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s2[0]; partial[2 * blockIdx.x + 1] = sp[0]; }
}

__global__ void finish_kernel(const double* __restrict__ partial, int nblk, int64_t n, double* __restrict__ out) {
    double s = 0, p = 0;
    for (int i = 0; i < nblk; ++i) { s += partial[2 * i]; p += partial[2 * i + 1]; }
    const double q = (double)n - p;
    out[0] = (p > 0 && q > 0) ? 0.5 * s / (p * q) : nan("");
    out[1] = p;
}

struct MaxOp {
    __host__ __device__ int64_t operator()(int64_t a, int64_t b) const { return a > b ? a : b; }
};

constexpr int NBLK = 1024;

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t cub_bytes(int64_t n) {
    size_t a = 0, b = 0, c = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const float*)nullptr, (float*)nullptr, (const uint8_t*)nullptr, (uint8_t*)nullptr, (int)n);
    (void)hipcub::DeviceScan::InclusiveScan(nullptr, b, (int64_t*)nullptr, (int64_t*)nullptr, MaxOp(), (int)n);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, c, (int64_t*)nullptr, (int64_t*)nullptr, (int)n);
    size_t m = a > b ? a : b;
    return m > c ? m : c;
}

}  // namespace

// bytes of device workspace ssad_auroc needs for n scores
extern "C" int64_t ssad_auroc_workspace(int64_t n) {
    if (n <= 0 || n >= (int64_t)2147483647) return -1;
    return (int64_t)(align256(cub_bytes(n)) + align256(n * 4) + align256(n) + 3 * align256(n * 8) + align256(NBLK * 16));
}

// scores fp32 [n], labels uint8 [n] (non-zero = positive) -> out[0] = AUROC (fp64), out[1] = number of positives
extern "C" int ssad_auroc(const float* scores, const uint8_t* labels, int64_t n, void* workspace, int64_t workspace_bytes,
                          double* out, void* stream) {
    SSAD_CHECK_ARG(scores && labels && workspace && out && n > 0 && n < (int64_t)2147483647, "bad argument");
    SSAD_CHECK_ARG(workspace_bytes >= ssad_auroc_workspace(n), "workspace too small (ssad_auroc_workspace)");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)workspace;
    size_t cb = cub_bytes(n);
    void* tmp = w; w += align256(cb);
    float* keys = (float*)w; w += align256(n * 4);
    uint8_t* lab = (uint8_t*)w; w += align256(n);
    int64_t* start = (int64_t*)w; w += align256(n * 8);
    int64_t* endr = (int64_t*)w; w += align256(n * 8);
    int64_t* cneg = (int64_t*)w; w += align256(n * 8);
    double* partial = (double*)w;
    size_t t = cb;
    (void)hipcub::DeviceRadixSort::SortPairs(tmp, t, scores, keys, labels, lab, (int)n, 0, 32, st);
    const unsigned g = (unsigned)cdiv64(n, 256);
    hipLaunchKernelGGL(mark_kernel, dim3(g), dim3(256), 0, st, keys, lab, n, start, cneg);
    hipLaunchKernelGGL(end_kernel, dim3(g), dim3(256), 0, st, keys, n, endr);
    t = cb; (void)hipcub::DeviceScan::InclusiveScan(tmp, t, start, start, MaxOp(), (int)n, st);
    t = cb; (void)hipcub::DeviceScan::InclusiveScan(tmp, t, endr, endr, MaxOp(), (int)n, st);
    t = cb; (void)hipcub::DeviceScan::InclusiveSum(tmp, t, cneg, cneg, (int)n, st);
    hipLaunchKernelGGL(contrib_kernel, dim3(NBLK), dim3(256), 0, st, lab, start, endr, cneg, n, partial);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(1), 0, st, partial, NBLK, n, out);
    SSAD_CHECK_LAUNCH();
    return 0;
}
