// Area under the ROC curve on the GPU: hand-written LSD radix sort + scans + tie-aware rank sum.
//
// Replaces sklearn.metrics.roc_curve + auc as the reference calls them for pixel / image AUROC
// (src/self_supervised/metrics.py:49-56, src/self_supervised/tools.py:76-98) when the scores already live on the GPU
// (83 x 65 536 pixel scores for one MVTec category).  The trapezoidal ROC area equals the Mann-Whitney statistic
// with ties counted one half:
//     AUC = sum over positives i of ( #negatives with a smaller score + 0.5 * #negatives with an equal score ) / (P * N)
// computed exactly in integers/halves and reduced in fp64 in a fixed order (deterministic).
//
// Sort (round 3: no library): scores -> order-preserving uint32 keys; four stable passes over 8-bit digits, each = per-block digit
// histogram (LDS integer atomics: counts are order-free), one exclusive scan of the [digit][block] table, and a scatter in which a
// block walks its 4 096 keys in rounds of 256 (index order) and ranks every key among the EARLIER keys of its digit: inside a wave by
// eight ballots (the lanes that agree on all eight digit bits), across the four waves and the rounds by running counters in LDS.
// HBM-bound byte work: 4 passes x (read keys + labels twice, write once) = ~60 bytes per score.
#include "common.h"

namespace {

constexpr int SORT_T = 256, SORT_ITEMS = 16, SORT_TILE = SORT_T * SORT_ITEMS;      // keys per workgroup

__device__ __forceinline__ uint32_t key_of(float f) {
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);        // ascending uint32 == ascending float (-0 sorts before +0)
}
__device__ __forceinline__ float float_of(uint32_t k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

__global__ void to_keys_kernel(const float* __restrict__ s, uint32_t* __restrict__ k, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) k[i] = key_of(s[i]);
}

// bh[d * nblk + b] = number of keys of block b whose digit (key >> shift) & 255 is d
__global__ __launch_bounds__(SORT_T) void digit_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int shift, int nblk,
                                                            uint32_t* __restrict__ bh) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = base + r * SORT_T + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    bh[(int64_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of `len` uint32 counters in place, one workgroup of 1024 threads (len = 256 * blocks: a few hundred thousand)
__global__ __launch_bounds__(1024) void scan_u32_kernel(uint32_t* __restrict__ a, int64_t len) {
    __shared__ uint32_t part[1024];
    const int64_t per = (len + 1023) / 1024;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < len ? lo + per : len;
    uint32_t s = 0;
    for (int64_t i = lo; i < hi; ++i) s += a[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 1024; ++t) { const uint32_t v = part[t]; part[t] = run; run += v; }
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (int64_t i = lo; i < hi; ++i) { const uint32_t v = a[i]; a[i] = run; run += v; }
}

// stable scatter of one digit: out position = goff[d][block] + (number of earlier keys of the block with digit d)
__global__ __launch_bounds__(SORT_T) void digit_scatter_kernel(const uint32_t* __restrict__ keys, const uint8_t* __restrict__ lab,
                                                               uint32_t* __restrict__ okeys, uint8_t* __restrict__ olab, int64_t n,
                                                               int shift, int nblk, const uint32_t* __restrict__ goff) {
    __shared__ uint32_t run[256];           // keys of digit d placed so far by this block (+ the block's global offset)
    __shared__ uint32_t wcnt[4][256];       // this round: keys of digit d in wave w
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    run[tid] = goff[(int64_t)tid * nblk + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = base + r * SORT_T + tid;
        const bool live = i < n;
        const uint32_t k = live ? keys[i] : 0u;
        const uint8_t l = live ? lab[i] : (uint8_t)0;
        const uint32_t d = (k >> shift) & 255u;
        // lanes of this wave with the same digit: agree on every one of its eight bits (dead lanes agree with nobody)
        uint64_t peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot(live && ((d >> b) & 1u));
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (live && rank == 0) wcnt[wave][d] = (uint32_t)__popcll(peers);        // the first lane of the group reports its size
        __syncthreads();
        if (live) {
            uint32_t pos = run[d] + rank;
            for (int w = 0; w < wave; ++w) pos += wcnt[w][d];
            okeys[pos] = k;
            olab[pos] = l;
        }
        __syncthreads();
        run[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
#pragma unroll
        for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0;
        __syncthreads();
    }
}

// ---- int64 inclusive scans (sum / running maximum): block-local scan + block totals, scan of the totals, add ----
struct SumOp { static __device__ __forceinline__ int64_t id() { return 0; } static __device__ __forceinline__ int64_t f(int64_t a, int64_t b) { return a + b; } };
struct MaxOp { static __device__ __forceinline__ int64_t id() { return INT64_MIN; } static __device__ __forceinline__ int64_t f(int64_t a, int64_t b) { return a > b ? a : b; } };

template <class Op>
__global__ __launch_bounds__(SORT_T) void scan_local_kernel(int64_t* __restrict__ a, int64_t n, int64_t* __restrict__ totals) {
    __shared__ int64_t part[SORT_T];
    const int64_t lo = (int64_t)blockIdx.x * SORT_TILE + (int64_t)threadIdx.x * SORT_ITEMS;       // a thread owns 16 consecutive entries
    int64_t v[SORT_ITEMS];
    int64_t s = Op::id();
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        v[j] = lo + j < n ? a[lo + j] : Op::id();
        s = Op::f(s, v[j]);
        v[j] = s;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = Op::id();
        for (int t = 0; t < SORT_T; ++t) { const int64_t x = part[t]; part[t] = run; run = Op::f(run, x); }
        totals[blockIdx.x] = run;
    }
    __syncthreads();
    const int64_t pre = part[threadIdx.x];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j)
        if (lo + j < n) a[lo + j] = Op::f(pre, v[j]);
}

template <class Op>
__global__ void scan_totals_kernel(int64_t* __restrict__ totals, int nblk) {      // exclusive, one thread: a few thousand entries
    if (threadIdx.x || blockIdx.x) return;
    int64_t run = Op::id();
    for (int b = 0; b < nblk; ++b) { const int64_t x = totals[b]; totals[b] = run; run = Op::f(run, x); }
}

template <class Op>
__global__ __launch_bounds__(SORT_T) void scan_add_kernel(int64_t* __restrict__ a, int64_t n, const int64_t* __restrict__ totals) {
    const int64_t pre = totals[blockIdx.x];
    const int64_t lo = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = lo + r * SORT_T + threadIdx.x;
        if (i < n) a[i] = Op::f(pre, a[i]);
    }
}

template <class Op>
void inclusive_scan(int64_t* a, int64_t n, int64_t* totals, int nblk, hipStream_t st) {
    hipLaunchKernelGGL(scan_local_kernel<Op>, dim3(nblk), dim3(SORT_T), 0, st, a, n, totals);
    hipLaunchKernelGGL(scan_totals_kernel<Op>, dim3(1), dim3(64), 0, st, totals, nblk);
    hipLaunchKernelGGL(scan_add_kernel<Op>, dim3(nblk), dim3(SORT_T), 0, st, a, n, totals);
}

// after sorting by score ascending: start[i] = index of the first element of i's tie group, cneg = inclusive scan of negatives
__global__ void mark_kernel(const uint32_t* __restrict__ keys, const uint8_t* __restrict__ lab, int64_t n, int64_t* __restrict__ start,
                            int64_t* __restrict__ neg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    start[i] = (i == 0 || float_of(keys[i]) != float_of(keys[i - 1])) ? i : 0;     // max-scan turns this into the group start
    neg[i] = lab[i] ? 0 : 1;
}

__global__ void end_kernel(const uint32_t* __restrict__ keys, int64_t n, int64_t* __restrict__ endm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // i enumerates the REVERSED array
    if (i >= n) return;
    const int64_t j = n - 1 - i;
    endm[i] = (j == n - 1 || float_of(keys[j]) != float_of(keys[j + 1])) ? -j : -(n + 1);     // min over the suffix as max of negatives
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

constexpr int NBLK = 1024;

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

// bytes of device workspace ssad_auroc needs for n scores
extern "C" int64_t ssad_auroc_workspace(int64_t n) {
    if (n <= 0 || n >= (int64_t)2147483647) return -1;
    const int64_t nblk = cdiv64(n, SORT_TILE);
    return (int64_t)(2 * align256(n * 4) + 2 * align256(n) + align256(256 * nblk * 4) + 3 * align256(n * 8) + align256(nblk * 8) +
                     align256(NBLK * 16));
}

// scores fp32 [n], labels uint8 [n] (non-zero = positive) -> out[0] = AUROC (fp64), out[1] = number of positives
extern "C" int ssad_auroc(const float* scores, const uint8_t* labels, int64_t n, void* workspace, int64_t workspace_bytes,
                          double* out, void* stream) {
    SSAD_CHECK_ARG(scores && labels && workspace && out && n > 0 && n < (int64_t)2147483647, "bad argument");
    SSAD_CHECK_ARG(workspace_bytes >= ssad_auroc_workspace(n), "workspace too small (ssad_auroc_workspace)");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)cdiv64(n, SORT_TILE);
    char* w = (char*)workspace;
    uint32_t* kA = (uint32_t*)w; w += align256(n * 4);
    uint32_t* kB = (uint32_t*)w; w += align256(n * 4);
    uint8_t* lA = (uint8_t*)w; w += align256(n);
    uint8_t* lB = (uint8_t*)w; w += align256(n);
    uint32_t* bh = (uint32_t*)w; w += align256((size_t)256 * nblk * 4);
    int64_t* start = (int64_t*)w; w += align256(n * 8);
    int64_t* endr = (int64_t*)w; w += align256(n * 8);
    int64_t* cneg = (int64_t*)w; w += align256(n * 8);
    int64_t* totals = (int64_t*)w; w += align256((size_t)nblk * 8);
    double* partial = (double*)w;
    const unsigned g = (unsigned)cdiv64(n, 256);
    hipLaunchKernelGGL(to_keys_kernel, dim3(g), dim3(256), 0, st, scores, kA, n);
    const uint8_t* lin = labels;
    for (int pass = 0; pass < 4; ++pass) {
        const uint32_t* kin = pass & 1 ? kB : kA;
        uint32_t* kout = pass & 1 ? kA : kB;
        uint8_t* lout = pass & 1 ? lA : lB;
        hipLaunchKernelGGL(digit_hist_kernel, dim3(nblk), dim3(SORT_T), 0, st, kin, n, 8 * pass, nblk, bh);
        hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(1024), 0, st, bh, (int64_t)256 * nblk);
        hipLaunchKernelGGL(digit_scatter_kernel, dim3(nblk), dim3(SORT_T), 0, st, kin, lin, kout, lout, n, 8 * pass, nblk, bh);
        lin = lout;
    }
    const uint32_t* keys = kA;           // four passes: the result is back in A
    const uint8_t* lab = lA;
    hipLaunchKernelGGL(mark_kernel, dim3(g), dim3(256), 0, st, keys, lab, n, start, cneg);
    hipLaunchKernelGGL(end_kernel, dim3(g), dim3(256), 0, st, keys, n, endr);
    inclusive_scan<MaxOp>(start, n, totals, nblk, st);
    inclusive_scan<MaxOp>(endr, n, totals, nblk, st);
    inclusive_scan<SumOp>(cneg, n, totals, nblk, st);
    hipLaunchKernelGGL(contrib_kernel, dim3(NBLK), dim3(256), 0, st, lab, start, endr, cneg, n, partial);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(1), 0, st, partial, NBLK, n, out);
    SSAD_CHECK_LAUNCH();
    return 0;
}
