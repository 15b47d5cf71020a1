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

// stable scatter of one digit: out position = goff[d][block] + (number of earlier keys of the block with digit d).
// P = the payload that travels with a key: the uint8 label (AUROC, F1 threshold) or the uint32 index of the score (PRO curve);
// lab == nullptr: the payload is the element's own index (first pass of an index sort)
template <class P>
__global__ __launch_bounds__(SORT_T) void digit_scatter_kernel(const uint32_t* __restrict__ keys, const P* __restrict__ lab,
                                                               uint32_t* __restrict__ okeys, P* __restrict__ olab, int64_t n,
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
        const P l = live ? (lab ? lab[i] : (P)i) : (P)0;
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
struct SumOp { typedef int64_t T; static __device__ __forceinline__ int64_t id() { return 0; } static __device__ __forceinline__ int64_t f(int64_t a, int64_t b) { return a + b; } };
struct MaxOp { typedef int64_t T; static __device__ __forceinline__ int64_t id() { return INT64_MIN; } static __device__ __forceinline__ int64_t f(int64_t a, int64_t b) { return a > b ? a : b; } };
// fp64 running sum (PRO curve): thread-serial over 16 entries, threads and blocks added in index order -- a fixed order (deterministic),
// not numpy's strictly sequential one: the two differ by rounding only (~1e-13 relative on 6 M terms)
struct SumF64 { typedef double T; static __device__ __forceinline__ double id() { return 0.0; } static __device__ __forceinline__ double f(double a, double b) { return a + b; } };

template <class Op>
__global__ __launch_bounds__(SORT_T) void scan_local_kernel(typename Op::T* __restrict__ a, int64_t n, typename Op::T* __restrict__ totals) {
    typedef typename Op::T T;
    __shared__ T part[SORT_T];
    const int64_t lo = (int64_t)blockIdx.x * SORT_TILE + (int64_t)threadIdx.x * SORT_ITEMS;       // a thread owns 16 consecutive entries
    T v[SORT_ITEMS];
    T s = Op::id();
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        v[j] = lo + j < n ? a[lo + j] : Op::id();
        s = Op::f(s, v[j]);
        v[j] = s;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = Op::id();
        for (int t = 0; t < SORT_T; ++t) { const T x = part[t]; part[t] = run; run = Op::f(run, x); }
        totals[blockIdx.x] = run;
    }
    __syncthreads();
    const T pre = part[threadIdx.x];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j)
        if (lo + j < n) a[lo + j] = Op::f(pre, v[j]);
}

template <class Op>
__global__ void scan_totals_kernel(typename Op::T* __restrict__ totals, int nblk) {      // exclusive, one thread: a few thousand entries
    if (threadIdx.x || blockIdx.x) return;
    typename Op::T run = Op::id();
    for (int b = 0; b < nblk; ++b) { const typename Op::T x = totals[b]; totals[b] = run; run = Op::f(run, x); }
}

template <class Op>
__global__ __launch_bounds__(SORT_T) void scan_add_kernel(typename Op::T* __restrict__ a, int64_t n, const typename Op::T* __restrict__ totals) {
    const typename Op::T pre = totals[blockIdx.x];
    const int64_t lo = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = lo + r * SORT_T + threadIdx.x;
        if (i < n) a[i] = Op::f(pre, a[i]);
    }
}

template <class Op>
void inclusive_scan(typename Op::T* a, int64_t n, typename Op::T* totals, int nblk, hipStream_t st) {
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
        hipLaunchKernelGGL(digit_scatter_kernel<uint8_t>, dim3(nblk), dim3(SORT_T), 0, st, kin, lin, kout, lout, n, 8 * pass, nblk, bh);
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

// =============================================================================================
// Round 4: the other sort-bound metrics of an evaluation on device-resident maps -- the MVTec PRO curve (metrics.compute_pro,
// src/self_supervised/metrics.py:58-190 of the reference), the F1-optimal threshold over the precision-recall curve
// (tools.Evaluator._get_threshold, tools.py:141-146: torchmetrics' PrecisionRecallCurve) and the confusion counts behind F1 / IoU
// (tools.py:131-139).  On the host these are three argsorts of 83 x 65 536 scores: 9 s per category, more than the training of
// that category takes on this card.  Same radix sort as above (descending here: complemented keys), payload = the score's index
// (PRO: its weights are gathered afterwards) or its label (F1).
// =============================================================================================
namespace {

__global__ void to_keys_desc_kernel(const float* __restrict__ s, uint32_t* __restrict__ k, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) k[i] = ~key_of(s[i]);
}
__device__ __forceinline__ float score_of_desc(uint32_t k) { return float_of(~k); }

// four stable passes; the result is back in (kA, pA).  first == nullptr: the payload is the element's index
template <class P>
void radix_sort_pairs(uint32_t* kA, uint32_t* kB, P* pA, P* pB, const P* first, int64_t n, int nblk, uint32_t* bh, hipStream_t st) {
    const P* pin = first;
    for (int pass = 0; pass < 4; ++pass) {
        const uint32_t* kin = pass & 1 ? kB : kA;
        uint32_t* kout = pass & 1 ? kA : kB;
        P* pout = pass & 1 ? pA : pB;
        hipLaunchKernelGGL(digit_hist_kernel, dim3(nblk), dim3(SORT_T), 0, st, kin, n, 8 * pass, nblk, bh);
        hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(1024), 0, st, bh, (int64_t)256 * nblk);
        hipLaunchKernelGGL(digit_scatter_kernel<P>, dim3(nblk), dim3(SORT_T), 0, st, kin, pin, kout, pout, n, 8 * pass, nblk, bh);
        pin = pout;
    }
}

// PRO: weights in sorted order + "last element of a run of equal scores" flags
__global__ void pro_gather_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ idx, const uint8_t* __restrict__ fp_w,
                                  const double* __restrict__ pro_w, int64_t n, int64_t* __restrict__ cfp, double* __restrict__ cpro,
                                  int64_t* __restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = idx[i];
    cfp[i] = fp_w[j] ? 1 : 0;
    cpro[i] = pro_w[j];
    keep[i] = (i == n - 1 || score_of_desc(keys[i]) != score_of_desc(keys[i + 1])) ? 1 : 0;
}

__global__ void pro_compact_kernel(const int64_t* __restrict__ cfp, const double* __restrict__ cpro, const int64_t* __restrict__ pos,
                                   int64_t n, double n_ok, double n_regions, float* __restrict__ fprs, double* __restrict__ pros,
                                   int64_t* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t before = i ? pos[i - 1] : 0;
    if (pos[i] != before) {                                   // this element closes a run: one curve point per distinct threshold
        const float f = (float)((double)cfp[i] / n_ok);       // numpy: float64 division, then astype(float32), then clip
        const double p = cpro[i] / n_regions;
        fprs[before] = f > 1.0f ? 1.0f : f;
        pros[before] = p > 1.0 ? 1.0 : p;
    }
    if (i == n - 1) count[0] = pos[i];
}

// F1 threshold: tps = inclusive count of positives in descending-score order, run ends, the first full-recall run end
__global__ void f1_mark_kernel(const uint32_t* __restrict__ keys, const uint8_t* __restrict__ lab, int64_t n, int64_t* __restrict__ tps,
                               uint8_t* __restrict__ run_end) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tps[i] = lab[i] ? 1 : 0;
    run_end[i] = (i == n - 1 || score_of_desc(keys[i]) != score_of_desc(keys[i + 1])) ? 1 : 0;
}

__global__ void f1_last_kernel(const int64_t* __restrict__ tps, const uint8_t* __restrict__ run_end, int64_t n,
                               unsigned long long* __restrict__ last) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (run_end[i] && tps[i] == tps[n - 1]) atomicMin(last, (unsigned long long)i);       // a minimum: order-free
}

// torchmetrics' float32 arithmetic (PrecisionRecallCurve of 0.8-0.10, then tools.py:145's 2PR / (P + R + 1e-10)); among equal F1 the
// SMALLEST threshold wins (np.argmax over ascending thresholds takes the first), i.e. the largest position in descending order
__device__ __forceinline__ float f1_at(int64_t i, int64_t tp, int64_t total) {
    const float tpsf = (float)tp;
    const float fpsf = __fsub_rn((float)(1 + i), tpsf);
    const float precision = __fdiv_rn(tpsf, __fadd_rn(tpsf, fpsf));
    const float recall = __fdiv_rn(tpsf, (float)total);
    return __fdiv_rn(__fmul_rn(__fmul_rn(2.f, precision), recall), __fadd_rn(__fadd_rn(precision, recall), 1e-10f));
}

__global__ __launch_bounds__(256) void f1_argmax_kernel(const int64_t* __restrict__ tps, const uint8_t* __restrict__ run_end, int64_t n,
                                                        const unsigned long long* __restrict__ last, float* __restrict__ pf,
                                                        int64_t* __restrict__ pi) {
    __shared__ float sf[256];
    __shared__ int64_t si[256];
    const int64_t lim = (int64_t)last[0], total = tps[n - 1];
    float bf = -1.f;
    int64_t bi = -1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= lim && i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!run_end[i]) continue;
        const float f = f1_at(i, tps[i], total);
        if (f > bf || (f == bf && i > bi)) { bf = f; bi = i; }
    }
    sf[threadIdx.x] = bf;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float f2 = sf[threadIdx.x + o];
            const int64_t i2 = si[threadIdx.x + o];
            if (f2 > sf[threadIdx.x] || (f2 == sf[threadIdx.x] && i2 > si[threadIdx.x])) { sf[threadIdx.x] = f2; si[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { pf[blockIdx.x] = sf[0]; pi[blockIdx.x] = si[0]; }
}

__global__ void f1_finish_kernel(const float* __restrict__ pf, const int64_t* __restrict__ pi, int nblk, const uint32_t* __restrict__ keys,
                                 const int64_t* __restrict__ tps, int64_t n, float* __restrict__ out) {
    float bf = -1.f;
    int64_t bi = -1;
    for (int b = 0; b < nblk; ++b)
        if (pf[b] > bf || (pf[b] == bf && pi[b] > bi)) { bf = pf[b]; bi = pi[b]; }
    if (tps[n - 1] == 0 || bi < 0) { bi = 0; bf = nanf(""); }          // no positive at all: every F1 is 0 / 0, argmax = the first
    out[0] = score_of_desc(keys[bi]);
    out[1] = bf;
}

__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ s, const uint8_t* __restrict__ t, int64_t n, float thr,
                                                        unsigned long long* __restrict__ out) {
    __shared__ unsigned long long sh[4][256];
    unsigned long long c[4] = {0, 0, 0, 0};               // tp, fp, fn, tn
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool p = s[i] >= thr, y = t[i] != 0;
        c[p ? (y ? 0 : 1) : (y ? 2 : 3)] += 1;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = c[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 4) atomicAdd(&out[threadIdx.x], sh[threadIdx.x][0]);       // integer sums: order-free
}

}  // namespace

extern "C" int64_t ssad_pro_curve_workspace(int64_t n) {
    if (n <= 0 || n >= (int64_t)2147483647) return -1;
    const int64_t nblk = cdiv64(n, SORT_TILE);
    return (int64_t)(4 * align256(n * 4) + align256(256 * nblk * 4) + 3 * align256(n * 8) + 2 * align256(nblk * 8));
}

// MVTec PRO curve of device-resident scores.  scores fp32 [n]; fp_w uint8 [n] (1 = defect-free pixel); pro_w fp64 [n] (1 / size of the
// pixel's ground-truth region, 0 outside regions); n_ok = number of defect-free pixels, n_regions = number of regions (both >= 1).
// -> fprs fp32 [count], pros fp64 [count] (one point per distinct score, descending; the caller adds (0, 0) and (1, 1)), count[0].
extern "C" int ssad_pro_curve(const float* scores, const uint8_t* fp_w, const double* pro_w, int64_t n, double n_ok, double n_regions,
                              void* workspace, int64_t workspace_bytes, float* fprs, double* pros, int64_t* count, void* stream) {
    SSAD_CHECK_ARG(scores && fp_w && pro_w && workspace && fprs && pros && count && n > 0 && n < (int64_t)2147483647, "bad argument");
    SSAD_CHECK_ARG(n_ok >= 1.0 && n_regions >= 1.0, "n_ok / n_regions must be at least 1 (numpy's max(., 1))");
    SSAD_CHECK_ARG(workspace_bytes >= ssad_pro_curve_workspace(n), "workspace too small (ssad_pro_curve_workspace)");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)cdiv64(n, SORT_TILE);
    char* w = (char*)workspace;
    uint32_t* kA = (uint32_t*)w; w += align256(n * 4);
    uint32_t* kB = (uint32_t*)w; w += align256(n * 4);
    uint32_t* iA = (uint32_t*)w; w += align256(n * 4);
    uint32_t* iB = (uint32_t*)w; w += align256(n * 4);
    uint32_t* bh = (uint32_t*)w; w += align256((size_t)256 * nblk * 4);
    int64_t* cfp = (int64_t*)w; w += align256(n * 8);
    double* cpro = (double*)w; w += align256(n * 8);
    int64_t* keep = (int64_t*)w; w += align256(n * 8);
    int64_t* totals = (int64_t*)w; w += align256((size_t)nblk * 8);
    double* dtotals = (double*)w;
    const unsigned g = (unsigned)cdiv64(n, 256);
    hipLaunchKernelGGL(to_keys_desc_kernel, dim3(g), dim3(256), 0, st, scores, kA, n);
    radix_sort_pairs<uint32_t>(kA, kB, iA, iB, nullptr, n, nblk, bh, st);
    hipLaunchKernelGGL(pro_gather_kernel, dim3(g), dim3(256), 0, st, kA, iA, fp_w, pro_w, n, cfp, cpro, keep);
    inclusive_scan<SumOp>(cfp, n, totals, nblk, st);
    inclusive_scan<SumF64>(cpro, n, dtotals, nblk, st);
    inclusive_scan<SumOp>(keep, n, totals, nblk, st);
    hipLaunchKernelGGL(pro_compact_kernel, dim3(g), dim3(256), 0, st, cfp, cpro, keep, n, n_ok, n_regions, fprs, pros, count);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t ssad_best_f1_workspace(int64_t n) {
    if (n <= 0 || n >= (int64_t)2147483647) return -1;
    const int64_t nblk = cdiv64(n, SORT_TILE);
    return (int64_t)(2 * align256(n * 4) + 3 * align256(n) + align256(256 * nblk * 4) + align256(n * 8) + align256(nblk * 8) +
                     align256(NBLK * 16) + 256);
}

// The threshold that maximises F1 over the precision-recall curve of (scores, targets == 1), as torchmetrics builds the curve
// (tools.py:141-146).  out[0] = threshold, out[1] = its F1 (float32 arithmetic of the reference's stack).
extern "C" int ssad_best_f1_threshold(const float* scores, const uint8_t* targets, int64_t n, void* workspace, int64_t workspace_bytes,
                                      float* out, void* stream) {
    SSAD_CHECK_ARG(scores && targets && workspace && out && n > 0 && n < (int64_t)2147483647, "bad argument");
    SSAD_CHECK_ARG(workspace_bytes >= ssad_best_f1_workspace(n), "workspace too small (ssad_best_f1_workspace)");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)cdiv64(n, SORT_TILE);
    char* w = (char*)workspace;
    uint32_t* kA = (uint32_t*)w; w += align256(n * 4);
    uint32_t* kB = (uint32_t*)w; w += align256(n * 4);
    uint8_t* lA = (uint8_t*)w; w += align256(n);
    uint8_t* lB = (uint8_t*)w; w += align256(n);
    uint8_t* run_end = (uint8_t*)w; w += align256(n);
    uint32_t* bh = (uint32_t*)w; w += align256((size_t)256 * nblk * 4);
    int64_t* tps = (int64_t*)w; w += align256(n * 8);
    int64_t* totals = (int64_t*)w; w += align256((size_t)nblk * 8);
    float* pf = (float*)w; w += align256(NBLK * 4);
    int64_t* pi = (int64_t*)w; w += align256(NBLK * 8);
    unsigned long long* last = (unsigned long long*)w;
    const unsigned g = (unsigned)cdiv64(n, 256);
    hipLaunchKernelGGL(to_keys_desc_kernel, dim3(g), dim3(256), 0, st, scores, kA, n);
    radix_sort_pairs<uint8_t>(kA, kB, lA, lB, targets, n, nblk, bh, st);
    hipLaunchKernelGGL(f1_mark_kernel, dim3(g), dim3(256), 0, st, kA, lA, n, tps, run_end);
    inclusive_scan<SumOp>(tps, n, totals, nblk, st);
    if (hipMemsetAsync(last, 0xff, 8, st) != hipSuccess) { ssad_set_error("ssad_best_f1_threshold: memset failed"); return 1; }
    hipLaunchKernelGGL(f1_last_kernel, dim3(g), dim3(256), 0, st, tps, run_end, n, last);
    hipLaunchKernelGGL(f1_argmax_kernel, dim3(NBLK), dim3(256), 0, st, tps, run_end, n, last, pf, pi);
    hipLaunchKernelGGL(f1_finish_kernel, dim3(1), dim3(1), 0, st, pf, pi, NBLK, kA, tps, n, out);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// Confusion counts of (scores >= threshold) against (targets != 0): out[0..3] = tp, fp, fn, tn (int64; the caller zeroes them).
extern "C" int ssad_confusion_counts(const float* scores, const uint8_t* targets, int64_t n, float threshold, int64_t* out, void* stream) {
    SSAD_CHECK_ARG(scores && targets && out && n > 0, "bad argument");
    hipLaunchKernelGGL(confusion_kernel, dim3(NBLK), dim3(256), 0, (hipStream_t)stream, scores, targets, n, threshold,
                       (unsigned long long*)out);
    SSAD_CHECK_LAUNCH();
    return 0;
}
