// Linear layers over a few hundred rows: the projection head and the classifier of a TRAINING batch
// (src/self_supervised/models.py:65-99 built, :247-252 applied; their gradients under loss.backward(), tools.py:270, :303).
//
// Why a kernel of its own (round 3, profiles/r03_b32_trace.md): with M = 32 .. 256 rows the implicit-GEMM tiles leave a grid of
// 8 - 16 workgroups that walk K = 512 .. 896 serially -- 22 us per 512 x 512 layer, 46 us for the 512 -> 4 classifier, 0.5 ms of a
// 6 ms batch-32 step for 0.06 % of its FLOPs.  Here the output is cut into 32 x 32 tiles, one workgroup each, and the CONTRACTION
// is dealt over the workgroup's waves (eight for forward / dgrad: 8-float chunks, wave w takes chunks w, w+8, ...; four for
// wgrad): every lane streams 16-byte pieces of its own A row and B row straight into v_mfma_f32_32x32x2_f32 operands (no LDS staging: lane half h supplies
// k = 8c + 4h + e for the e-th MFMA of chunk c, the same permutation on both operands), eight chunks of loads in flight, and
// the partial tiles are added in wave order through LDS (a fixed order: the result does not depend on scheduling).
//
//   y[M][N] = a[M][K] . b[N][K]^T   (* scale[n] + shift[n]) (+ residual) (relu)        forward / input gradient
//   dw[O][K] (+)= dz[M][O]^T . x[M][K]                                                   weight gradient
//
// Train-mode BatchNorm1d statistics are left as per-row-tile double partial sums, the format of ssad_conv_igemm_fwd_stats.
#include "common.h"
#include <stdlib.h>

namespace {

struct SmallGemm {
    const float* a;
    const float* b;
    float* y;
    const float* scale;
    const float* shift;
    const float* residual;
    double* stats;            // [row tiles][2][N] or null
    int M, K, N, relu;
};

constexpr int RP = 33;        // LDS row pitch of a 32 x 32 partial tile

// the wave's 32 x 32 partial (MFMA accumulator layout of common.h) -> red[wave][row][col]
__device__ __forceinline__ void spill_tile(float* red, const f32x16& acc, int lane) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int v = 0; v < 16; ++v) red[((v & 3) + 8 * (v >> 2) + 4 * h) * RP + j] = acc[v];
}

constexpr int GW = 8;         // waves of a forward / input-gradient workgroup: K = 512 is one batch of eight chunks per wave

// RND: operands rounded to bf16 (1) / fp16 (2) as they are loaded -- the precision-16 step's linear layers (autocast runs them on 16-bit
// operands with fp32 accumulation; products of two 16-bit values are exact in fp32, so the fp32 MFMA over rounded operands is that
// arithmetic in another summation order).  Round 5: these layers ran on the 16-bit implicit GEMM before, 21-28 us per launch against 7.5.
template <int RND> __device__ __forceinline__ float rnd16(float x) { return x; }
template <> __device__ __forceinline__ float rnd16<1>(float x) { return (float)(__bf16)x; }
template <> __device__ __forceinline__ float rnd16<2>(float x) { return (float)(hf)x; }

template <int RND>
__global__ __launch_bounds__(64 * GW) void small_gemm_kernel(SmallGemm p) {
    __shared__ float red[GW][32 * RP];
    __shared__ float cs[2][2 * GW][32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const bool aok = m0 + r < p.M, bok = n0 + r < p.N;
    const float* ap = p.a + (size_t)(aok ? m0 + r : 0) * p.K + 4 * h;
    const float* bp = p.b + (size_t)(bok ? n0 + r : 0) * p.K + 4 * h;
    const int nchunk = (p.K + 7) / 8;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    constexpr int U = 8;
    for (int c = w; c < nchunk; c += GW * U) {
        f32x4 av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c + GW * u;
            const bool ok = cc < nchunk && cc * 8 + 4 * h < p.K;       // K % 4 == 0: a 16-byte piece is inside or outside
            av[u] = ok && aok ? *(const f32x4*)(ap + (size_t)cc * 8) : zero;
            bv[u] = ok && bok ? *(const f32x4*)(bp + (size_t)cc * 8) : zero;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma32(rnd16<RND>(av[u][e]), rnd16<RND>(bv[u][e]), acc);
    }
    spill_tile(red[w], acc, lane);
    __syncthreads();
    // thread (j, ig): column j, rows ig and ig + 16; the eight partial tiles are added in wave order
    const int j = tid & 31, ig = tid >> 5;
    const int n = n0 + j;
    const float sc = (p.scale && n < p.N) ? p.scale[n] : 1.f;
    const float sh = (p.shift && n < p.N) ? p.shift[n] : 0.f;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 32 / (2 * GW); ++q) {
        const int i = ig + 2 * GW * q, m = m0 + i;
        float v = red[0][i * RP + j];
#pragma unroll
        for (int ww = 1; ww < GW; ++ww) v += red[ww][i * RP + j];
        if (m < p.M && n < p.N) {
            s0 += v;
            s1 += v * v;
            v = v * sc + sh;
            if (p.residual) v += p.residual[(size_t)m * p.N + n];
            if (p.relu) v = fmaxf(v, 0.f);
            p.y[(size_t)m * p.N + n] = v;
        }
    }
    if (p.stats) {
        cs[0][ig][j] = s0;
        cs[1][ig][j] = s1;
        __syncthreads();
        if (tid < 64) {
            const int which = tid >> 5;
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 2 * GW; ++g) t += cs[which][g][j];
            if (n < p.N) p.stats[((size_t)blockIdx.y * 2 + which) * p.N + n] = (double)t;
        }
    }
}

// dw[O][K] (+)= sum_m dz[m][o] * x[m][k]: lane (r, h) of a pair step supplies dz[2q + h][o0 + r] and x[2q + h][k0 + r] -- two
// 128-byte rows per operand per MFMA; wave w takes the row pairs q = w, w + 4, ...
template <int RND>
__global__ __launch_bounds__(256) void small_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                           float* __restrict__ dw, int M, int O, int K, int accumulate) {
    __shared__ float red[4][32 * RP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int k0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    const bool aok = o0 + r < O, bok = k0 + r < K;
    const int npair = (M + 1) / 2;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    constexpr int U = 8;
    for (int q = w; q < npair; q += 4 * U) {
        float av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // branch-free: every lane loads from a valid (clamped) address and the value is selected afterwards.  With the loads inside
            // `ok ? load : 0` branches hipcc sank the 16-bit rounding of the RND != 0 instantiations INTO each branch -- load, s_waitcnt
            // vmcnt(0), two converts, sixteen times in a row: 17-22 us per launch where the exact-fp32 instantiation takes 6-8
            // (round 6, found in profiles/r06_p16_b256_trace_step.csv)
            const int m = 2 * (q + 4 * u) + h;
            const bool ok = m < M;
            const int mc = ok ? m : 0;
            const float ta = dz[(size_t)mc * O + (aok ? o0 + r : 0)];
            const float tb = x[(size_t)mc * K + (bok ? k0 + r : 0)];
            av[u] = ok && aok ? ta : 0.f;
            bv[u] = ok && bok ? tb : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = mfma32(rnd16<RND>(av[u]), rnd16<RND>(bv[u]), acc);
    }
    spill_tile(red[w], acc, lane);
    __syncthreads();
    const int j = tid & 31, ig = tid >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = ig + 8 * q;
        if (o0 + i < O && k0 + j < K) {
            float v = ((red[0][i * RP + j] + red[1][i * RP + j]) + red[2][i * RP + j]) + red[3][i * RP + j];
            float* d = dw + (size_t)(o0 + i) * K + k0 + j;
            *d = accumulate ? *d + v : v;
        }
    }
}

int max_rows() {
    static const int v = getenv("SSAD_LINEAR_SMALL") ? atoi(getenv("SSAD_LINEAR_SMALL")) : 512;
    return v;
}

}  // namespace

// Largest row count the small-batch linear kernels take (0: switched off with SSAD_LINEAR_SMALL=0).
extern "C" int ssad_linear_small_max_rows(void) { return max_rows(); }

// conv_igemm.hip's forward / input-gradient entry points hand 1 x 1 layers over 1 x 1 maps with few rows to this kernel.
bool ssad_linear_small_ok(const void* a, const void* b, int64_t M, int K) {
    return M > 0 && M <= max_rows() && K % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0;
}

int ssad_linear_small_launch(const float* a, const float* b, float* y, const float* scale, const float* shift,
                             const float* residual, int relu, int M, int K, int N, double* stats, int* stat_rows, void* stream, int round) {
    SmallGemm p{a, b, y, scale, shift, residual, stats, M, K, N, relu};
    dim3 grid((N + 31) / 32, (M + 31) / 32);
    if (round == 2) hipLaunchKernelGGL(small_gemm_kernel<2>, grid, dim3(64 * GW), 0, (hipStream_t)stream, p);
    else if (round == 1) hipLaunchKernelGGL(small_gemm_kernel<1>, grid, dim3(64 * GW), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(small_gemm_kernel<0>, grid, dim3(64 * GW), 0, (hipStream_t)stream, p);
    if (stat_rows) *stat_rows = (int)grid.y;
    SSAD_CHECK_LAUNCH();
    return 0;
}

// Weight gradient of a linear layer over M <= ssad_linear_small_max_rows() rows: dw[Cout][Cin] (+)= dy[M][Cout]^T x[M][Cin].
// round: 0 exact fp32; 1 / 2: operands rounded to bf16 / fp16 while loaded (the precision-16 step's linear layers).
extern "C" int ssad_linear_wgrad_small_r(const float* dy, const float* x, float* dw, int64_t M, int Cin, int Cout, int accumulate,
                                         int round, void* stream) {
    SSAD_CHECK_ARG(dy && x && dw, "null pointer");
    SSAD_CHECK_ARG(M > 0 && M <= max_rows() && Cin > 0 && Cout > 0, "row count outside the small-batch range");
    SSAD_CHECK_ARG(round >= 0 && round <= 2, "round: 0 (fp32), 1 (bf16 operands) or 2 (fp16 operands)");
    dim3 grid((Cin + 31) / 32, (Cout + 31) / 32);
    hipStream_t st = (hipStream_t)stream;
    if (round == 2) hipLaunchKernelGGL(small_wgrad_kernel<2>, grid, dim3(256), 0, st, dy, x, dw, (int)M, Cout, Cin, accumulate);
    else if (round == 1) hipLaunchKernelGGL(small_wgrad_kernel<1>, grid, dim3(256), 0, st, dy, x, dw, (int)M, Cout, Cin, accumulate);
    else hipLaunchKernelGGL(small_wgrad_kernel<0>, grid, dim3(256), 0, st, dy, x, dw, (int)M, Cout, Cin, accumulate);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_linear_wgrad_small(const float* dy, const float* x, float* dw, int64_t M, int Cin, int Cout, int accumulate,
                                       void* stream) {
    return ssad_linear_wgrad_small_r(dy, x, dw, M, Cin, Cout, accumulate, 0, stream);
}
