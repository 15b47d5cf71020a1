// HBM-bound kernels of the training step: train-mode BatchNorm (forward statistics / apply, backward reduce /
// apply), max-pool and global-average-pool backward, stem im2col, softmax cross-entropy, fused SGD, weight
// flip-transpose for dgrad.
//
// Replaces the BatchNorm2d/1d, MaxPool2d, adaptive_avg_pool2d, F.cross_entropy autograd nodes and
// torch.optim.SGD(momentum=0.9, weight_decay=5e-4) of the reference's training step
// (src/self_supervised/models.py:256-277, :336-341).
// Per-channel sums accumulate in fp64 and are combined in a fixed order (two-stage, no atomics), so batch
// statistics and BN gradients are deterministic and free of E[x^2]-E[x]^2 cancellation.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// ---------------------------------------------------------------------------------------------
// column reductions over [R][C] (C % 4 == 0): threads = TC channel quads x RL row lanes
// ---------------------------------------------------------------------------------------------
struct ColReduce {
    int TC, RL;
};
static inline ColReduce col_geom(int C, int E = 4) {
    // at most 16 lanes (256 bytes) along a row: tensors of many channels and few rows (layer3 / layer4) then spread over gx column
    // blocks x nblk row blocks instead of nblk workgroups of two row lanes with 8+ dependent load rounds each
    int TC = C / E;
    if (TC > 16) TC = 16;
    while (256 % TC) --TC;
    return {TC, 256 / TC};
}

// Nibble masks: one byte per channel quad, bit k = "the ReLU output of channel 4 q + k was positive".  A lane owns E = 4 (float: one
// byte) or 8 (half: two bytes, read / written as one uint16) consecutive channels; pass bit of the lane's channel k:
template <int E> __device__ __forceinline__ unsigned mask_word(const uint8_t* __restrict__ m, int64_t i) {
    if (E == 4) return m[i];
    return ((const uint16_t*)m)[i];
}
template <int E> __device__ __forceinline__ bool mask_bit(unsigned w, int k) { return (w >> (E == 4 || k < 4 ? k : k + 4)) & 1u; }

// mode 0: s0 = sum z, s1 = sum z^2
// mode 1: g = dy * (yact > 0 if yact) ; s0 = sum g ; s1 = sum g * (z - mean) * invstd   (z may be null -> s1 = 0)
//         mask source: yact (the saved activation) or, when zmask_gamma is given (no residual fed the ReLU), the
//         sign of (z - mean) * invstd * gamma + beta recomputed from z -- one tensor less to read
//         or mask4: one byte per channel quad, bit k = "the ReLU output of channel 4q+k was positive" (written by the forward
//         BatchNorm apply: 1/16 of the bytes of the activation it stands for; fp32 tensors only)
// A lane owns E = 4 (float) or 8 (half) consecutive channels: one 16-byte access per tensor and row.
template <int MODE, typename T = float>
__global__ __launch_bounds__(256) void col_reduce_kernel(const T* __restrict__ a, const T* __restrict__ yact, const T* __restrict__ z,
                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                  const float* __restrict__ zmask_gamma, const float* __restrict__ zmask_beta,
                                  double* __restrict__ partial, int64_t R, int C, int TC, int RL, int rows_per_block,
                                  const uint8_t* __restrict__ mask4 = nullptr) {
    constexpr int E = Lane<T>::E;
    using V = typename Lane<T>::vec;
    __shared__ double sh[2][256][E];
    const int tid = threadIdx.x;
    const int tx = tid % TC, ty = tid / TC;
    const int CE = C / E;
    for (int cq = blockIdx.x * TC + tx; cq < CE && ty < RL; cq += gridDim.x * TC) {
        double s0[E], s1[E];
#pragma unroll
        for (int k = 0; k < E; ++k) s0[k] = s1[k] = 0;
        V mu = 0.f, is = 1.f, mg = 0.f, mb = 0.f;
        if (MODE == 1 && z) { mu = ldpar<V>(mean, cq); is = ldpar<V>(invstd, cq); }
        if (MODE == 1 && zmask_gamma) { mg = ldpar<V>(zmask_gamma, cq); mb = ldpar<V>(zmask_beta, cq); }
        const int64_t rb = (int64_t)blockIdx.y * rows_per_block;
        const int64_t re = rb + rows_per_block < R ? rb + rows_per_block : R;
        // U rows per iteration (64 bytes per tensor in flight per lane): all loads of an iteration are issued before any is consumed
        // (latency-bound otherwise).  Measured on the first half instantiation (round 5, 8-byte accesses, precision-16 step at batch
        // 256: 19 launches, 1.0 ms, 2.5 TB/s): 8 rows per iteration 4.8 ms (the row arrays leave the registers); the rows of an
        // iteration summed in fp32 before they enter the double accumulators -- a quarter of the fp64 instructions -- 0.99 ms: unchanged.
        constexpr int U = 16 / E;
        for (int64_t row = rb + ty; row < re; row += (int64_t)U * RL) {
            V v[U], ya[U], zz[U];
            typename RawLane<T>::t rv[U], rya[U], rzz[U];
            unsigned mk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t rw = row + (int64_t)u * RL;
                const int64_t o = (rw < re ? rw : rb + ty) * CE + cq;       // clamp: tail rows re-read a valid row ...
                rv[u] = ldrawv(a + E * o);                                  // raw pieces: the conversions follow all the requests
                if (MODE == 1 && yact) rya[u] = ldrawv(yact + E * o);
                if (MODE == 1 && mask4) mk[u] = mask_word<E>(mask4, o);
                if (MODE == 1 && z) rzz[u] = ldrawv(z + E * o);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] = cvtraw(rv[u]);
                if (MODE == 1 && yact) ya[u] = cvtraw(rya[u]);
                if (MODE == 1 && z) zz[u] = cvtraw(rzz[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (row + (int64_t)u * RL >= re) continue;                  // ... and are dropped here
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < E; ++k) { s0[k] += (double)v[u][k]; s1[k] += (double)v[u][k] * (double)v[u][k]; }
                } else {
                    if (yact) {
#pragma unroll
                        for (int k = 0; k < E; ++k) v[u][k] = ya[u][k] > 0.f ? v[u][k] : 0.f;
                    } else if (mask4) {
#pragma unroll
                        for (int k = 0; k < E; ++k) v[u][k] = mask_bit<E>(mk[u], k) ? v[u][k] : 0.f;
                    } else if (zmask_gamma) {
#pragma unroll
                        for (int k = 0; k < E; ++k) v[u][k] = (zz[u][k] - mu[k]) * is[k] * mg[k] + mb[k] > 0.f ? v[u][k] : 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < E; ++k) s0[k] += (double)v[u][k];
                    if (z) {
#pragma unroll
                        for (int k = 0; k < E; ++k) s1[k] += (double)v[u][k] * (double)((zz[u][k] - mu[k]) * is[k]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < E; ++k) { sh[0][tid][k] = s0[k]; sh[1][tid][k] = s1[k]; }
    }
    __syncthreads();
    if (ty == 0) {
        for (int cq = blockIdx.x * TC + tx; cq < CE; cq += gridDim.x * TC) {
            double s0[E], s1[E];
#pragma unroll
            for (int k = 0; k < E; ++k) s0[k] = s1[k] = 0;
            for (int l = 0; l < RL; ++l)
#pragma unroll
                for (int k = 0; k < E; ++k) { s0[k] += sh[0][l * TC + tx][k]; s1[k] += sh[1][l * TC + tx][k]; }
            double* pp = partial + (int64_t)blockIdx.y * 2 * C;
#pragma unroll
            for (int k = 0; k < E; ++k) { pp[cq * E + k] = s0[k]; pp[C + cq * E + k] = s1[k]; }
        }
    }
}

// Sum of the per-block partials of one channel by one 256-thread workgroup: thread t adds partials t, t+256, ... in
// order, a fixed xor-butterfly combines the 64 lanes of a wave and thread 0 adds the four wave sums in order
// (deterministic).  The conv epilogue leaves up to M/128 partial rows, so a single wave per channel would spend
// tens of microseconds in dependent loads.
__device__ __forceinline__ bool channel_totals(const double* __restrict__ partial, int nblk, int C, int c, double& s0, double& s1) {
    __shared__ double wsum[4][2];
    double a = 0, b = 0;
    int k = threadIdx.x;
    for (; k + 7 * 256 < nblk; k += 8 * 256) {         // eight rows in flight per thread, added in the same order as one by one
        double pa[8], pb[8];                            // (the halo-tile convs leave up to M / 128 = 8 192 partial rows at batch 256)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            pa[u] = partial[(int64_t)(k + u * 256) * 2 * C + c];
            pb[u] = partial[(int64_t)(k + u * 256) * 2 * C + C + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a += pa[u]; b += pb[u]; }
    }
    for (; k < nblk; k += 256) {
        a += partial[(int64_t)k * 2 * C + c];
        b += partial[(int64_t)k * 2 * C + C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6][0] = a; wsum[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    s0 = ((wsum[0][0] + wsum[1][0]) + wsum[2][0]) + wsum[3][0];
    s1 = ((wsum[0][1] + wsum[1][1]) + wsum[2][1]) + wsum[3][1];
    return threadIdx.x == 0;
}

// BN forward statistics: mean, invstd (biased var), running stats (unbiased var, momentum).  One workgroup per channel.
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const double* __restrict__ partial, int nblk, int64_t R, int C, float eps,
                                         float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                         float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x;
    double s0, s1;
    if (!channel_totals(partial, nblk, C, c, s0, s1)) return;
    const double m = s0 / (double)R;
    double var = s1 / (double)R - m * m;
    if (var < 0) var = 0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblk, int C, float* __restrict__ dbeta,
                                       float* __restrict__ dgamma) {
    const int c = blockIdx.x;
    double s0, s1;
    if (!channel_totals(partial, nblk, C, c, s0, s1)) return;
    if (dbeta) dbeta[c] = (float)s0;
    if (dgamma) dgamma[c] = (float)s1;
}

// y = (z - mean) * invstd * gamma + beta (+ res) (relu).  A lane owns E = 4 (float) / 8 (half) consecutive channels; totalE, CE in
// units of E.  mask4 (one byte per channel quad): fp32 tensors only.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const T* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const T* __restrict__ res, T* __restrict__ y, int64_t totalE, int CE, int relu,
                                    uint8_t* __restrict__ mask4 = nullptr) {
    constexpr int E = Lane<T>::E;
    using V = typename Lane<T>::vec;
    // A thread's channel group is the same in every iteration whenever the grid's stride is a multiple of the groups per row (always for
    // the trunk's 64 .. 512 channels): its four parameter vectors are then loaded ONCE, not once per 16 bytes of tensor (round 6: over
    // halves the per-iteration form issued 128 bytes of parameter loads per 48 bytes of data).
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    const bool fixed = gstride % CE == 0;
    const int cq0 = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) % CE);
    V mu = ldpar<V>(mean, cq0), is = ldpar<V>(invstd, cq0);
    V g = ldpar<V>(gamma, cq0), b = ldpar<V>(beta, cq0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < totalE; i += gstride) {
        if (!fixed) {
            const int cq = (int)(i % CE);
            mu = ldpar<V>(mean, cq); is = ldpar<V>(invstd, cq);
            g = ldpar<V>(gamma, cq); b = ldpar<V>(beta, cq);
        }
        // (both requests before either conversion: see ldrawv in common.h)
        const typename RawLane<T>::t rz = ldrawv(z + E * i);
        typename RawLane<T>::t rres = rz;
        if (res) rres = ldrawv(res + E * i);
        V v = cvtraw(rz);
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = (v[k] - mu[k]) * is[k] * g[k] + b[k];
        if (res) {
            const V rr = cvtraw(rres);
#pragma unroll
            for (int k = 0; k < E; ++k) v[k] += rr[k];
        }
        if (mask4) {
            // (halves: the mask is that of the STORED activation -- a positive value that rounds to zero as a half would pass a
            // gradient the stored tensor's ReLU does not; rounding to nearest keeps the sign, so only underflow to +0 differs)
            unsigned mw = 0;
#pragma unroll
            for (int k = 0; k < E; ++k) mw |= (unsigned)(stored<T>(v[k]) > 0.f) << (E == 4 || k < 4 ? k : k + 4);
            if (E == 4) mask4[i] = (uint8_t)mw;
            else ((uint16_t*)mask4)[i] = (uint16_t)mw;
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < E; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        stv(y + E * i, v);
    }
}

// g = dy * (yact > 0); dz = gamma*invstd*(g - dbeta/R - xhat*dgamma/R)  [train]   or  g*gamma*invstd  [eval];
// optionally dres = g (gradient of the identity branch of a residual block)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ yact, const T* __restrict__ z,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ dbeta,
                                    const float* __restrict__ dgamma, T* __restrict__ dz, T* __restrict__ dres,
                                    int64_t totalE, int CE, float invR, int eval_mode, const float* __restrict__ zmask_beta,
                                    const uint8_t* __restrict__ mask4 = nullptr) {
    constexpr int E = Lane<T>::E;
    using V = typename Lane<T>::vec;
    // (parameters once per thread when the grid's stride keeps its channel group fixed: see bn_apply_fwd_kernel)
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    const bool fixed = gstride % CE == 0;
    int cq = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) % CE);
    V mu = ldpar<V>(mean, cq), is = ldpar<V>(invstd, cq), ga = ldpar<V>(gamma, cq);
    V zb_f = 0.f, db_f = 0.f, dg_f = 0.f;
    if (zmask_beta) zb_f = ldpar<V>(zmask_beta, cq);
    if (!eval_mode) { db_f = ldpar<V>(dbeta, cq); dg_f = ldpar<V>(dgamma, cq); }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < totalE; i += gstride) {
        if (!fixed) {
            cq = (int)(i % CE);
            mu = ldpar<V>(mean, cq); is = ldpar<V>(invstd, cq); ga = ldpar<V>(gamma, cq);
            if (zmask_beta) zb_f = ldpar<V>(zmask_beta, cq);
            if (!eval_mode) { db_f = ldpar<V>(dbeta, cq); dg_f = ldpar<V>(dgamma, cq); }
        }
        // every tensor this element needs is requested before anything is converted or used (ldrawv, common.h): dy, then the saved
        // activation or the mask, then z (mask from z and / or the train-mode apply)
        const typename RawLane<T>::t rg = ldrawv(dy + E * i);
        typename RawLane<T>::t rya = rg, rzr = rg;
        unsigned mkw = 0;
        const bool use_ya = yact != nullptr, use_mk = !use_ya && mask4 != nullptr, use_zm = !use_ya && !use_mk && zmask_beta != nullptr;
        if (use_ya) rya = ldrawv(yact + E * i);
        if (use_mk) mkw = mask_word<E>(mask4, i);
        if (use_zm || !eval_mode) rzr = ldrawv(z + E * i);
        V g = cvtraw(rg);
        const V zall = cvtraw(rzr);
        if (use_ya) {
            const V ya = cvtraw(rya);
#pragma unroll
            for (int k = 0; k < E; ++k) g[k] = ya[k] > 0.f ? g[k] : 0.f;
        } else if (use_mk) {
            const unsigned mk = mkw;
#pragma unroll
            for (int k = 0; k < E; ++k) g[k] = mask_bit<E>(mk, k) ? g[k] : 0.f;
        } else if (use_zm) {                     // ReLU mask recomputed from z (layer without residual)
            const V zb = zb_f;
            const V zm = zall;
#pragma unroll
            for (int k = 0; k < E; ++k) g[k] = (zm[k] - mu[k]) * is[k] * ga[k] + zb[k] > 0.f ? g[k] : 0.f;
        }
        if (dres) stv(dres + E * i, g);
        V o;
        if (eval_mode) {
#pragma unroll
            for (int k = 0; k < E; ++k) o[k] = g[k] * ga[k] * is[k];
        } else {
            const V db = db_f, dg = dg_f;
            const V zz = zall;
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const float xh = (zz[k] - mu[k]) * is[k];
                o[k] = ga[k] * is[k] * (g[k] - db[k] * invR - xh * dg[k] * invR);
            }
        }
        stv(dz + E * i, o);
    }
}

// ---------------------------------------------------------------------------------------------
// pooling backward
// ---------------------------------------------------------------------------------------------
// dx[n][y][x][c] = sum over the (<= 4) 3x3/2 windows covering (y,x) whose FIRST maximum (row-major scan, as
// PyTorch's max_pool2d picks it) is (y,x).  Gather form: deterministic, no atomics.
__global__ void maxpool_bwd_kernel(const float* __restrict__ xin, const float* __restrict__ dy, float* __restrict__ dx,
                                   int64_t total4, int H, int W, int C4, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i % C4);
    int64_t t = i / C4;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int64_t n = t / H;
    const f32x4* xp = (const f32x4*)xin + n * H * W * C4 + c4;
    const f32x4 self = xp[((int64_t)y * W + x) * C4];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oy0 = y / 2, oy1 = (y + 1) / 2, ox0 = x / 2, ox1 = (x + 1) / 2;     // windows with 2*o-1 <= pos <= 2*o+1
    for (int oy = oy0; oy <= oy1; ++oy) {
        if (oy >= Ho) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            if (ox >= Wo) continue;
            // (y,x) is the window's FIRST maximum (PyTorch's tie rule) iff every earlier position (row-major) is
            // strictly smaller and every later one is not larger
            bool win[4] = {true, true, true, true};
#pragma unroll
            for (int dyy = 0; dyy < 3; ++dyy) {
                const int yy = 2 * oy - 1 + dyy;
                if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
                for (int dxx = 0; dxx < 3; ++dxx) {
                    const int xx = 2 * ox - 1 + dxx;
                    if ((unsigned)xx >= (unsigned)W || (yy == y && xx == x)) continue;
                    const f32x4 v = xp[((int64_t)yy * W + xx) * C4];
                    const bool earlier = yy < y || (yy == y && xx < x);
#pragma unroll
                    for (int k = 0; k < 4; ++k) win[k] = win[k] && (earlier ? v[k] < self[k] : v[k] <= self[k]);
                }
            }
            const f32x4 g = ((const f32x4*)dy)[((n * Ho + oy) * Wo + ox) * C4 + c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += win[k] ? g[k] : 0.f;
        }
    }
    ((f32x4*)dx)[i] = acc;
}

// Index form: idx[n][oy][ox][c] = window slot (dy*3+dx) of the first maximum, written by the forward kernel.
// dx[n][y][x][c] = sum over the <= 4 windows covering (y,x) whose recorded slot is (y,x).
__global__ void maxpool_bwd_idx_kernel(const uint8_t* __restrict__ idx, const float* __restrict__ dy, float* __restrict__ dx,
                                       int64_t total4, int H, int W, int C4, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i % C4);
    int64_t t = i / C4;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int64_t n = t / H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oy0 = y / 2, oy1 = (y + 1) / 2, ox0 = x / 2, ox1 = (x + 1) / 2;
    for (int oy = oy0; oy <= oy1; ++oy) {
        if (oy >= Ho) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            if (ox >= Wo) continue;
            const uint32_t slot = (uint32_t)((y - (2 * oy - 1)) * 3 + (x - (2 * ox - 1)));
            const int64_t o = ((n * Ho + oy) * Wo + ox) * C4 + c4;
            const uint32_t packed = ((const uint32_t*)idx)[o];
            const f32x4 g = ((const f32x4*)dy)[o];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += ((packed >> (8 * k)) & 0xffu) == slot ? g[k] : 0.f;
        }
    }
    ((f32x4*)dx)[i] = acc;
}

// ---- stem head of the backward pass, fused: max-pool backward (index form) feeding BatchNorm + ReLU backward ----
// g[n][y][x][c] = (sum of the pooled gradients whose argmax slot is (y,x)) * [relu mask recomputed from z]; the 1 GB
// gradient map between the two never exists in HBM.
__device__ __forceinline__ f32x4 pool_grad(const uint8_t* __restrict__ idx, const float* __restrict__ dpool, int64_t n, int y,
                                           int x, int c4, int C4, int Ho, int Wo) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oy0 = y / 2, oy1 = (y + 1) / 2, ox0 = x / 2, ox1 = (x + 1) / 2;
    for (int oy = oy0; oy <= oy1; ++oy) {
        if (oy >= Ho) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            if (ox >= Wo) continue;
            const uint32_t slot = (uint32_t)((y - (2 * oy - 1)) * 3 + (x - (2 * ox - 1)));
            const int64_t o = ((n * Ho + oy) * Wo + ox) * C4 + c4;
            const uint32_t packed = ((const uint32_t*)idx)[o];
            const f32x4 g = ((const f32x4*)dpool)[o];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += ((packed >> (8 * k)) & 0xffu) == slot ? g[k] : 0.f;
        }
    }
    return acc;
}

// APPLY == 0: per-block partial sums  s0 = sum g, s1 = sum g * xhat  (double, [gridDim.x][2][C])
// APPLY == 1: dz = gamma * invstd * (g - dbeta / R - xhat * dgamma / R)
// A thread works on a 2 x 2 block of pixels: the four pooled cells (2 x 2) that can have their argmax inside the block are read
// ONCE (index word + gradient quad each) and dealt to the four pixels by slot, instead of four cell look-ups per pixel -- the
// look-ups, not the streaming of z, set the pace of the one-pixel form (1.19 ms per step at 3.2 TB/s -> see DESIGN.md).
// Block (a, b) = pixels (2a .. 2a+1, 2b .. 2b+1); cell (oy, ox) covers pixels 2oy-1 .. 2oy+1, so the block's pixels are covered by
// cells oy in {a, a+1}, ox in {b, b+1} only; slot of pixel (y, x) in cell (oy, ox) = (y - 2oy + 1) * 3 + (x - 2ox + 1).
// A lane owns E = 4 (float) / 8 (half) consecutive channels (16-byte accesses; E / 4 index words per cell).
template <int APPLY, typename T = float>
__global__ __launch_bounds__(256) void pool_bn_bwd_kernel(const uint8_t* __restrict__ idx, const T* __restrict__ dpool,
                                                          const T* __restrict__ z, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ dbeta,
                                                          const float* __restrict__ dgamma, T* __restrict__ dz,
                                                          double* __restrict__ partial, int64_t R, int H, int W, int C,
                                                          int Ho, int Wo, int64_t rows_per_block) {
    constexpr int E = Lane<T>::E;
    using V = typename Lane<T>::vec;
    __shared__ double sh[2][256][E];
    const int CE = C / E, RL = 256 / CE;
    const int tid = threadIdx.x, cq = tid % CE, ty = tid / CE;
    const V mu = ldpar<V>(mean, cq), is = ldpar<V>(invstd, cq);
    const V ga = ldpar<V>(gamma, cq), be = ldpar<V>(beta, cq);
    V db = 0.f, dg = 0.f;
    if (APPLY) { db = ldpar<V>(dbeta, cq); dg = ldpar<V>(dgamma, cq); }
    const float invR = 1.f / (float)R;
    double s0[E], s1[E];
#pragma unroll
    for (int k = 0; k < E; ++k) s0[k] = s1[k] = 0;
    // "rows" here are 2 x 2 pixel blocks: Hb x Wb per image (H, W even or odd: the last block row / column may be half empty)
    const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;
    const int64_t NB = (R / ((int64_t)H * W)) * Hb * Wb;
    const int64_t rb = (int64_t)blockIdx.x * rows_per_block;
    const int64_t re = rb + rows_per_block < NB ? rb + rows_per_block : NB;
    for (int64_t blk = rb + ty; blk < re; blk += RL) {
        const int b = (int)(blk % Wb);
        const int64_t t = blk / Wb;
        const int a = (int)(t % Hb);
        const int64_t n = t / Hb;
        // the four candidate cells
        uint32_t pk[2][2][E / 4];
        V gq[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int oy = a + dy, ox = b + dx;
                const bool ok = oy < Ho && ox < Wo;
                const int64_t o = ((n * Ho + (ok ? oy : 0)) * Wo + (ok ? ox : 0)) * CE + cq;
#pragma unroll
                for (int q = 0; q < E / 4; ++q) pk[dy][dx][q] = ok ? ((const uint32_t*)idx)[o * (E / 4) + q] : 0xffffffffu;   // slot 255 never matches
                gq[dy][dx] = ldv(dpool + E * o);
            }
        // ... and the block's four z values, requested with them (branch-free: a pixel outside the map reads the block's first pixel
        // and is skipped below) -- inside the per-pixel bounds branch each load was waited for on its own (round 6, ISA inspection)
        V zq[2][2];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int y = 2 * a + py, x = 2 * b + px;
                const bool ok = y < H && x < W;
                const int64_t rowc = (n * H + (ok ? y : 2 * a)) * W + (ok ? x : 2 * b);
                zq[py][px] = ldv(z + E * (rowc * CE + cq));
            }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int y = 2 * a + py, x = 2 * b + px;
                if (y >= H || x >= W) continue;
                V g = 0.f;
                // same cell order as the one-pixel form (oy ascending, then ox): cells oy in {y/2, (y+1)/2}, ox likewise
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int oy = a + dy, ox = b + dx;
                        if (dy > py || dx > px) continue;       // pixel (py, px) of the block lies in cells a .. a + py, b .. b + px
                        const uint32_t slot = (uint32_t)((y - (2 * oy - 1)) * 3 + (x - (2 * ox - 1)));
#pragma unroll
                        for (int k = 0; k < E; ++k) g[k] += ((pk[dy][dx][k >> 2] >> (8 * (k & 3))) & 0xffu) == slot ? gq[dy][dx][k] : 0.f;
                    }
                const int64_t row = (n * H + y) * W + x;
                const V zz = zq[py][px];
                V xh;
#pragma unroll
                for (int k = 0; k < E; ++k) {
                    xh[k] = (zz[k] - mu[k]) * is[k];
                    g[k] = (zz[k] - mu[k]) * is[k] * ga[k] + be[k] > 0.f ? g[k] : 0.f;      // the forward's expression
                }
                if (APPLY) {
                    V o;
#pragma unroll
                    for (int k = 0; k < E; ++k) o[k] = ga[k] * is[k] * (g[k] - db[k] * invR - xh[k] * dg[k] * invR);
                    stv(dz + E * (row * CE + cq), o);
                } else {
#pragma unroll
                    for (int k = 0; k < E; ++k) { s0[k] += (double)g[k]; s1[k] += (double)g[k] * (double)xh[k]; }
                }
            }
    }
    if (APPLY) return;
#pragma unroll
    for (int k = 0; k < E; ++k) { sh[0][tid][k] = s0[k]; sh[1][tid][k] = s1[k]; }
    __syncthreads();
    if (ty == 0) {
        double a0[E], a1[E];
#pragma unroll
        for (int k = 0; k < E; ++k) a0[k] = a1[k] = 0;
        for (int l = 0; l < RL; ++l)
#pragma unroll
            for (int k = 0; k < E; ++k) { a0[k] += sh[0][l * CE + cq][k]; a1[k] += sh[1][l * CE + cq][k]; }
        double* pp = partial + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int k = 0; k < E; ++k) { pp[cq * E + k] = a0[k]; pp[C + cq * E + k] = a1[k]; }
    }
}

// dy[n][hw][c] (+)= dpooled[n*stride + off + c] / HW      (four channels per thread when C % 4 == 0: 16-byte accesses)
__global__ void gap_bwd_kernel(const float* __restrict__ dpooled, float* __restrict__ dy, int64_t total, int HW, int C,
                               int stride, int off, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int64_t n = i / ((int64_t)HW * C);
    const float g = dpooled[n * stride + off + c] / (float)HW;
    dy[i] = accumulate ? dy[i] + g : g;
}
template <typename T>
__global__ void gap_bwd4_kernel(const float* __restrict__ dpooled, T* __restrict__ dy, int64_t total4, int HW, int C4,
                                int stride, int off, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i % C4);
    const int64_t n = i / ((int64_t)HW * C4);
    const float* gp = dpooled + n * stride + off + c4 * 4;
    f32x4 g = {gp[0] / (float)HW, gp[1] / (float)HW, gp[2] / (float)HW, gp[3] / (float)HW};
    T* d = dy + 4 * i;
    st4(d, accumulate ? ld4(d) + g : g);
}

// ---------------------------------------------------------------------------------------------
// stem im2col: Xcol[m][k], k = (ky*7 + kx)*3 + c for k < 147, zero for 147 <= k < 160; nearest resize fused
// ---------------------------------------------------------------------------------------------
__global__ void stem_im2col_kernel(const float* __restrict__ img, float* __restrict__ col, int64_t total4, int H, int W,
                                   int Hv, int Wv, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte piece of a 160-float row
    if (i >= total4) return;
    const int k4 = (int)(i % 40);
    int64_t m = i / 40;
    const int ox = (int)(m % Wo); m /= Wo;
    const int oy = (int)(m % Ho);
    const int64_t n = m / Ho;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k4 * 4 + j;
        if (k < 147) {
            const int c = k % 3, tap = k / 3, ky = tap / 7, kx = tap % 7;
            const int vy = 2 * oy - 3 + ky, vx = 2 * ox - 3 + kx;
            if ((unsigned)vy < (unsigned)Hv && (unsigned)vx < (unsigned)Wv) {
                const int sy = (vy * H) / Hv, sx = (vx * W) / Wv;
                v[j] = img[((n * 3 + c) * H + sy) * W + sx];
            }
        }
    }
    ((f32x4*)col)[i] = v;
}

// OIHW [64][3][7][7] -> [64][160] rows in im2col k order (zero padded)
__global__ void pack_stem_weight_2d_kernel(const float* __restrict__ w, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 160) return;
    const int k = i % 160, co = i / 160;
    float v = 0.f;
    if (k < 147) {
        const int c = k % 3, tap = k / 3, ky = tap / 7, kx = tap % 7;
        v = w[((co * 3 + c) * 7 + ky) * 7 + kx];
    }
    out[i] = v;
}

// OHWI [O][KH][KW][I] -> [I][KH][KW][O] with both taps reversed (dgrad operand)
__global__ void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int I, int KH, int KW) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)O * I * KH * KW) return;
    const int o = (int)(idx % O);
    int64_t t = idx / O;
    const int kx = (int)(t % KW); t /= KW;
    const int ky = (int)(t % KH);
    const int i = (int)(t / KH);
    out[idx] = w[(((int64_t)o * KH + (KH - 1 - ky)) * KW + (KW - 1 - kx)) * I + i];
}

// The same for up to 32 filters of one flat arena in ONE launch (a training step flips every conv weight once: 19 launches
// of a few microseconds each otherwise).  tab[k] = {src offset, dst offset, O, I, KH, KW, first linear index}.
struct FlipTable {
    int n;
    int64_t e[32][7];        // {src offset, dst offset, O, I, KH, KW, first workgroup of this filter}
    int64_t total;           // workgroups
};
// One workgroup = one 32 x 32 (o, i) tile of one tap: read coalesced along i, written coalesced along o through LDS (the
// one-thread-per-element form read with a stride of O floats: 96 us per step for 45 MB, 0.9 TB/s).
template <typename TO>
__global__ __launch_bounds__(256) void flip_transpose_batch_kernel(const float* __restrict__ src, TO* __restrict__ dst, FlipTable t) {
    __shared__ float tile[32][33];
    int k = 0;
    while (k + 1 < t.n && (int64_t)blockIdx.x >= t.e[k + 1][6]) ++k;
    const int O = (int)t.e[k][2], I = (int)t.e[k][3], KH = (int)t.e[k][4], KW = (int)t.e[k][5];
    const int T = KH * KW, ot = (O + 31) / 32, it = (I + 31) / 32;
    int l = (int)((int64_t)blockIdx.x - t.e[k][6]);
    const int ti = l % it; l /= it;
    const int to = l % ot;
    const int tap = l / ot;                                   // destination tap (ky', kx'); the source tap is the mirrored one
    const float* w = src + t.e[k][0];
    TO* out = dst + t.e[k][1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = to * 32 + ty + 8 * r, i = ti * 32 + tx;
        tile[ty + 8 * r][tx] = (o < O && i < I) ? w[((int64_t)o * T + (T - 1 - tap)) * I + i] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = ti * 32 + ty + 8 * r, o = to * 32 + tx;
        if (o < O && i < I) out[((int64_t)i * T + tap) * O + o] = (TO)tile[tx][ty + 8 * r];
    }
}

// ---------------------------------------------------------------------------------------------
// softmax cross-entropy (mean over the batch) + accuracy + dlogits; one workgroup, fixed reduction order
// ---------------------------------------------------------------------------------------------
__global__ void softmax_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int B, int C,
                                  float* __restrict__ loss_acc, float* __restrict__ dlogits, int ldd, float gscale) {
    __shared__ double sl[256];
    __shared__ int sc[256];
    double l = 0;
    int correct = 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* p = logits + (int64_t)b * C;
        float mx = p[0];
        int am = 0;
        for (int c = 1; c < C; ++c)
            if (p[c] > mx) { mx = p[c]; am = c; }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
        const float lse = logf(se) + mx;
        const int y = (int)labels[b];
        l += (double)(lse - p[y]);
        correct += am == y;
        if (dlogits) {
            float* d = dlogits + (int64_t)b * ldd;
            for (int c = 0; c < ldd; ++c) d[c] = c < C ? (expf(p[c] - lse) - (c == y ? 1.f : 0.f)) * gscale : 0.f;
        }
    }
    sl[threadIdx.x] = l;
    sc[threadIdx.x] = correct;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tl = 0;
        int tc = 0;
        for (int i = 0; i < blockDim.x; ++i) { tl += sl[i]; tc += sc[i]; }
        loss_acc[0] = (float)(tl / B);
        loss_acc[1] = (float)tc / (float)B;
    }
}

// m = mu*m + (g*gscale + wd*p); p -= lr*m      (torch.optim.SGD, dampening 0, no nesterov; m starts at 0)
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, int64_t n, float lr,
                           float mu, float wd, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float d = g[i] * gscale + wd * pv;
        const float mv = mu * m[i] + d;
        m[i] = mv;
        p[i] = pv - lr * mv;
    }
}

// ---- device-resident optimiser state (so that a captured step, a hipGraph, never bakes in a learning rate or a loss
// scale): hyper = [lr, momentum, weight_decay, grad_scale]; scaler = [loss_scale, growth_tracker, found_inf] ----
// The fp16-operand path (Trainer(precision=16)) follows torch.cuda.amp.GradScaler as PL drives it for the reference
// (tools.py:263): loss * scale before backward, gradients unscaled in the update, the update skipped and the scale
// halved when a gradient is not finite, the scale doubled after `interval` clean steps.
__global__ void check_finite_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ scaler) {
    bool bad = false;
    // 16 bytes per lane, two pieces in flight (the arena ranges start at multiples of 32 bytes; the tail element-wise)
    const int64_t n4 = (((uintptr_t)g & 15) == 0) ? n / 4 : 0;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 2 * step) {
        const f32x4 a = ((const f32x4*)g)[i];
        const f32x4 b = ((const f32x4*)g)[i + step < n4 ? i + step : i];
#pragma unroll
        for (int k = 0; k < 4; ++k) bad = bad || !(fabsf(a[k]) <= 3.402823466e38f) || !(fabsf(b[k]) <= 3.402823466e38f);       // inf or nan
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        const float v = g[i];
        bad = bad || !(fabsf(v) <= 3.402823466e38f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) scaler[2] = 1.f;   // benign race: every writer stores the same value
}

__global__ void sgd_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, int64_t n,
                               const float* __restrict__ hyper, const float* __restrict__ scaler) {
    if (scaler && scaler[2] != 0.f) return;                // non-finite gradients: GradScaler skips optimizer.step()
    const float lr = hyper[0], mu = hyper[1], wd = hyper[2];
    const float gscale = scaler ? hyper[3] / scaler[0] : hyper[3];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float d = g[i] * gscale + wd * pv;
        const float mv = mu * m[i] + d;
        m[i] = mv;
        p[i] = pv - lr * mv;
    }
}

__global__ void scaler_update_kernel(float* __restrict__ scaler, float growth, float backoff, float interval) {
    if (threadIdx.x || blockIdx.x) return;
    if (scaler[2] != 0.f) {
        scaler[0] *= backoff;
        scaler[1] = 0.f;
    } else if (++scaler[1] >= interval) {
        scaler[0] *= growth;
        scaler[1] = 0.f;
    }
    scaler[2] = 0.f;
}

__global__ void scale_by_dev_kernel(float* __restrict__ x, int64_t n, const float* __restrict__ scaler) {
    const float s = scaler[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= s;
}

static inline unsigned ew_grid(int64_t n) {
    int64_t b = cdiv64(n, 256);
    return (unsigned)(b < 8192 ? (b < 1 ? 1 : b) : 8192);
}

}  // namespace

// Number of doubles of workspace the column reductions need for R rows x C channels.
// Row blocks of a column reduction over R rows: 32 row sweeps per workgroup on big tensors (<= 2048 blocks), but never fewer than
// ~256 workgroups while a block still has one full sweep of 4 rows per row lane -- the tensors of a batch-32 step (2 048 .. 131 072
// rows) used to get 32 .. 256 workgroups of eight dependent load rounds each: 11-15 us per reduction for 4-33 MB, now 5-8 (round 4:
// batch-32 step 5.59 -> 5.51 ms).
static inline int64_t col_blocks(int64_t R, const ColReduce& g) {
    int64_t nblk = cdiv64(R, (int64_t)g.RL * 32);
    const int64_t fine = cdiv64(R, (int64_t)g.RL * 4);
    const int64_t want = fine < 256 ? fine : 256;
    if (nblk < want) nblk = want;
    if (nblk > 2048) nblk = 2048;
    if (nblk < 1) nblk = 1;
    return nblk;
}

extern "C" int64_t ssad_colreduce_workspace(int64_t R, int C) {
    ColReduce g = col_geom(C);                        // the four-channel (float) geometry: never fewer row blocks than the half one
    return col_blocks(R, g) * 2 * C;
}

template <typename T>
static int launch_col_reduce(int mode, const T* a, const T* yact, const T* z, const float* mean,
                             const float* invstd, double* ws, int64_t R, int C, int* nblk_out, hipStream_t st,
                             const float* zg = nullptr, const float* zb = nullptr, const uint8_t* mask4 = nullptr) {
    constexpr int E = Lane<T>::E;
    ColReduce g = col_geom(C, E);
    int64_t nblk = col_blocks(R, g);
    int rows_per_block = (int)cdiv64(R, nblk);
    nblk = cdiv64(R, rows_per_block);                 // no empty trailing blocks (their partial rows would be read uninitialised)
    int gx = (C / E + g.TC - 1) / g.TC;
    dim3 grid(gx, (unsigned)nblk);
    if (mode == 0)
        hipLaunchKernelGGL((col_reduce_kernel<0, T>), grid, dim3(256), 0, st, a, yact, z, mean, invstd, zg, zb, ws, R, C, g.TC, g.RL, rows_per_block,
                           (const uint8_t*)nullptr);
    else
        hipLaunchKernelGGL((col_reduce_kernel<1, T>), grid, dim3(256), 0, st, a, yact, z, mean, invstd, zg, zb, ws, R, C, g.TC, g.RL, rows_per_block,
                           mask4);
    *nblk_out = (int)nblk;
    return 0;
}

int ssad_bn_finalize_partials(const double* partial, int nblk, int64_t R, int C, float eps, float momentum, float* mean,
                              float* invstd, float* running_mean, float* running_var, void* stream) {
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, nblk, R, C,
                       eps, momentum, mean, invstd, running_mean, running_var);
    SSAD_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int bn_stats_impl(const T* z, int64_t R, int C, float eps, float momentum, float* mean, float* invstd,
                         float* running_mean, float* running_var, double* workspace, void* stream) {
    SSAD_CHECK_ARG(z && mean && invstd && workspace && R > 0 && C > 0 && C % Lane<T>::E == 0, "bad argument");
    int nblk;
    launch_col_reduce<T>(0, z, nullptr, nullptr, nullptr, nullptr, workspace, R, C, &nblk, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, workspace, nblk, R, C,
                       eps, momentum, mean, invstd, running_mean, running_var);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_bn_stats(const float* z, int64_t R, int C, float eps, float momentum, float* mean, float* invstd,
                             float* running_mean, float* running_var, double* workspace, void* stream) {
    return bn_stats_impl<float>(z, R, C, eps, momentum, mean, invstd, running_mean, running_var, workspace, stream);
}

// ..._h: the same kernels over tensors stored as halves (arithmetic, statistics, parameters and their gradients stay fp32): the
// precision-16 training step, whose activations live in HBM as torch.autocast stores them (tools.py:263 of the reference)
extern "C" int ssad_bn_stats_h(const void* z, int64_t R, int C, float eps, float momentum, float* mean, float* invstd,
                               float* running_mean, float* running_var, double* workspace, void* stream) {
    return bn_stats_impl<hf>((const hf*)z, R, C, eps, momentum, mean, invstd, running_mean, running_var, workspace, stream);
}

template <typename T>
static int bn_apply_fwd_impl(const T* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             const T* residual, T* y, int64_t R, int C, int relu, void* stream) {
    constexpr int E = Lane<T>::E;
    SSAD_CHECK_ARG(z && mean && invstd && gamma && beta && y && R > 0 && C > 0 && C % E == 0, "bad argument");
    const int64_t totalE = R * (C / E);
    hipLaunchKernelGGL(bn_apply_fwd_kernel<T>, dim3(ew_grid(totalE)), dim3(256), 0, (hipStream_t)stream, z, mean, invstd, gamma,
                       beta, residual, y, totalE, C / E, relu, (uint8_t*)nullptr);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_bn_apply_fwd(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                 const float* residual, float* y, int64_t R, int C, int relu, void* stream) {
    return bn_apply_fwd_impl<float>(z, mean, invstd, gamma, beta, residual, y, R, C, relu, stream);
}

extern "C" int ssad_bn_apply_fwd_h(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                   const void* residual, void* y, int64_t R, int C, int relu, void* stream) {
    return bn_apply_fwd_impl<hf>((const hf*)z, mean, invstd, gamma, beta, (const hf*)residual, (hf*)y, R, C, relu, stream);
}

// The same apply, also leaving the ReLU's active set as a nibble mask (one byte per channel quad, bit k = output of channel
// 4q+k positive): the backward pass of a residual block reads 1/16 of the activation's bytes instead of the activation.
extern "C" int ssad_bn_apply_fwd_mask(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                      const float* residual, float* y, uint8_t* mask4, int64_t R, int C, int relu, void* stream) {
    SSAD_CHECK_ARG(z && mean && invstd && gamma && beta && y && mask4 && R > 0 && C > 0 && C % 4 == 0, "bad argument");
    const int64_t total4 = R * (C / 4);
    hipLaunchKernelGGL(bn_apply_fwd_kernel<float>, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, z, mean, invstd, gamma,
                       beta, residual, y, total4, C / 4, relu, mask4);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// ... over half tensors (round 6: the precision-16 step's residual blocks keep the mask as the fp32 step does: 1/8 of the bytes of the
// half activation the backward passes would otherwise read twice, and no materialised identity-branch gradient)
extern "C" int ssad_bn_apply_fwd_mask_h(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                        const void* residual, void* y, uint8_t* mask4, int64_t R, int C, int relu, void* stream) {
    SSAD_CHECK_ARG(z && mean && invstd && gamma && beta && y && mask4 && R > 0 && C > 0 && C % 8 == 0, "bad argument");
    const int64_t total8 = R * (C / 8);
    hipLaunchKernelGGL(bn_apply_fwd_kernel<hf>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream, (const hf*)z, mean, invstd, gamma,
                       beta, (const hf*)residual, (hf*)y, total8, C / 8, relu, mask4);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// Column sums of a tiny matrix (the bias gradient of the 512 -> 4 classifier over a training batch: 32 x 4 values) in ONE launch:
// the general two-stage reduction gives such a matrix one workgroup whose thread 0 adds 256 LDS values one after the other, then
// a finalize launch -- 13 + 4 us for 128 numbers.  Lane l of column c adds rows l, l + L, ...; eight lane groups are added in
// order, then the eight group sums: a fixed order, in double.
__global__ __launch_bounds__(256) void col_sum_tiny_kernel(const float* __restrict__ a, float* __restrict__ out, int R, int C) {
    __shared__ double sh[256];
    __shared__ double grp[8][32];
    const int c = threadIdx.x % C, l = threadIdx.x / C, L = 256 / C;        // C in {4, 8, 16, 32}: L = 64 .. 8
    double s = 0;
    for (int r = l; r < R; r += L) s += (double)a[(int64_t)r * C + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    const int per = L / 8;                                                   // lanes per group
    if (threadIdx.x < 8 * C) {
        const int g = threadIdx.x / C, cc = threadIdx.x % C;
        double t = 0;
        for (int k = 0; k < per; ++k) t += sh[(g * per + k) * C + cc];
        grp[g][cc] = t;
    }
    __syncthreads();
    if (threadIdx.x < C) {
        double t = grp[0][threadIdx.x];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += grp[g][threadIdx.x];
        out[threadIdx.x] = (float)t;
    }
}

// dbeta/dgamma over rows of g = dy*(yact>0); with z == NULL only dbeta (= column sums: Linear bias gradient).
template <typename T>
static int bn_bwd_reduce_impl(const T* dy, const T* yact, const T* z, const float* mean, const float* invstd,
                              float* dbeta, float* dgamma, int64_t R, int C, double* workspace, void* stream,
                              const float* zg, const float* zb, const uint8_t* mask4 = nullptr) {
    SSAD_CHECK_ARG(dy && workspace && R > 0 && C > 0 && C % Lane<T>::E == 0, "bad argument");
    SSAD_CHECK_ARG(!z || (mean && invstd), "z needs mean/invstd");
    SSAD_CHECK_ARG(!zg || (z && zb && !yact), "mask-from-z needs z, gamma, beta and no yact");
    if constexpr (std::is_same<T, float>::value) {
        if (!z && !yact && !mask4 && dbeta && R <= 4096 && (C == 4 || C == 8 || C == 16 || C == 32)) {     // plain column sums, tiny
            hipLaunchKernelGGL(col_sum_tiny_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dy, dbeta, (int)R, C);
            SSAD_CHECK_LAUNCH();
            return 0;
        }
    }
    int nblk;
    launch_col_reduce<T>(1, dy, yact, z, mean, invstd, workspace, R, C, &nblk, (hipStream_t)stream, zg, zb, mask4);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, workspace, nblk, C, dbeta,
                       z ? dgamma : nullptr);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_bn_bwd_reduce(const float* dy, const float* yact, const float* z, const float* mean, const float* invstd,
                                  float* dbeta, float* dgamma, int64_t R, int C, double* workspace, void* stream) {
    return bn_bwd_reduce_impl<float>(dy, yact, z, mean, invstd, dbeta, dgamma, R, C, workspace, stream, nullptr, nullptr);
}

extern "C" int ssad_bn_bwd_reduce_h(const void* dy, const void* yact, const void* z, const float* mean, const float* invstd,
                                    float* dbeta, float* dgamma, int64_t R, int C, double* workspace, void* stream) {
    return bn_bwd_reduce_impl<hf>((const hf*)dy, (const hf*)yact, (const hf*)z, mean, invstd, dbeta, dgamma, R, C, workspace, stream,
                                  nullptr, nullptr);
}

// Same reductions with the ReLU mask recomputed from z: mask = (z - mean) * invstd * gamma + beta > 0 (a BN + ReLU with no
// residual in between), so the saved activation does not have to be read again.
extern "C" int ssad_bn_bwd_reduce_zmask(const float* dy, const float* z, const float* mean, const float* invstd,
                                        const float* gamma, const float* beta, float* dbeta, float* dgamma, int64_t R, int C,
                                        double* workspace, void* stream) {
    SSAD_CHECK_ARG(gamma && beta, "null gamma/beta");
    return bn_bwd_reduce_impl<float>(dy, nullptr, z, mean, invstd, dbeta, dgamma, R, C, workspace, stream, gamma, beta);
}

extern "C" int ssad_bn_bwd_reduce_zmask_h(const void* dy, const void* z, const float* mean, const float* invstd,
                                          const float* gamma, const float* beta, float* dbeta, float* dgamma, int64_t R, int C,
                                          double* workspace, void* stream) {
    SSAD_CHECK_ARG(gamma && beta, "null gamma/beta");
    return bn_bwd_reduce_impl<hf>((const hf*)dy, nullptr, (const hf*)z, mean, invstd, dbeta, dgamma, R, C, workspace, stream, gamma, beta);
}

template <typename T>
static int bn_apply_bwd_impl(const T* dy, const T* yact, const T* z, const float* mean, const float* invstd,
                                 const float* gamma, const float* dbeta, const float* dgamma, T* dz, T* dres,
                                 int64_t R, int C, int eval_mode, void* stream, const float* zmask_beta,
                                 const uint8_t* mask4 = nullptr) {
    constexpr int E = Lane<T>::E;
    SSAD_CHECK_ARG(dy && mean && invstd && gamma && dz && R > 0 && C > 0 && C % E == 0, "bad argument");
    SSAD_CHECK_ARG(eval_mode || (z && dbeta && dgamma), "train-mode backward needs z, dbeta, dgamma");
    const int64_t totalE = R * (C / E);
    hipLaunchKernelGGL(bn_apply_bwd_kernel<T>, dim3(ew_grid(totalE)), dim3(256), 0, (hipStream_t)stream, dy, yact, z, mean, invstd,
                       gamma, dbeta, dgamma, dz, dres, totalE, C / E, 1.f / (float)R, eval_mode, zmask_beta, mask4);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_bn_apply_bwd(const float* dy, const float* yact, const float* z, const float* mean, const float* invstd,
                                 const float* gamma, const float* dbeta, const float* dgamma, float* dz, float* dres,
                                 int64_t R, int C, int eval_mode, void* stream) {
    return bn_apply_bwd_impl<float>(dy, yact, z, mean, invstd, gamma, dbeta, dgamma, dz, dres, R, C, eval_mode, stream, nullptr);
}

extern "C" int ssad_bn_apply_bwd_h(const void* dy, const void* yact, const void* z, const float* mean, const float* invstd,
                                   const float* gamma, const float* dbeta, const float* dgamma, void* dz, void* dres,
                                   int64_t R, int C, int eval_mode, void* stream) {
    return bn_apply_bwd_impl<hf>((const hf*)dy, (const hf*)yact, (const hf*)z, mean, invstd, gamma, dbeta, dgamma, (hf*)dz, (hf*)dres,
                                 R, C, eval_mode, stream, nullptr);
}

extern "C" int ssad_bn_apply_bwd_zmask(const float* dy, const float* z, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, const float* dbeta, const float* dgamma, float* dz,
                                       int64_t R, int C, void* stream) {
    SSAD_CHECK_ARG(z && beta, "mask-from-z needs z and beta");
    return bn_apply_bwd_impl<float>(dy, nullptr, z, mean, invstd, gamma, dbeta, dgamma, dz, nullptr, R, C, 0, stream, beta);
}

extern "C" int ssad_bn_apply_bwd_zmask_h(const void* dy, const void* z, const float* mean, const float* invstd,
                                         const float* gamma, const float* beta, const float* dbeta, const float* dgamma, void* dz,
                                         int64_t R, int C, void* stream) {
    SSAD_CHECK_ARG(z && beta, "mask-from-z needs z and beta");
    return bn_apply_bwd_impl<hf>((const hf*)dy, nullptr, (const hf*)z, mean, invstd, gamma, dbeta, dgamma, (hf*)dz, nullptr, R, C, 0,
                                 stream, beta);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm over a few hundred rows -- the BatchNorm1d layers of the projection head on a training batch
// (src/self_supervised/models.py:65-95) -- statistics, finalisation and apply in ONE launch each way (round 3: at batch 32 the
// three launches of the general path cost 15 us per layer for 64 KB of data, profiles/r03_b32_trace.md).  A workgroup owns 32
// channels for ALL rows: thread (j, ig) = channel j of the group, rows ig, ig + 8, ...; column sums in double per thread, the
// eight row groups added in order through LDS (deterministic); the rows are then read again (R x 128 bytes per workgroup: cache
// hits) for the apply.  Formulas are those of bn_stats_finalize_kernel / bn_apply_fwd_kernel / col_reduce_kernel<1> /
// bn_apply_bwd_kernel.
// ---------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ void small_col_totals(double (*sh)[8][32], int nsum, const double* part, int ig, int j, double* tot) {
    for (int k = 0; k < nsum; ++k) sh[k][ig][j] = part[k];
    __syncthreads();
    for (int k = 0; k < nsum; ++k) {
        double t = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += sh[k][g][j];
        tot[k] = t;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const float* __restrict__ z, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ invstd,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var, int R,
                                                            int C, float eps, float momentum, int relu) {
    __shared__ double sh[2][8][32];
    const int j = threadIdx.x & 31, ig = threadIdx.x >> 5, c = blockIdx.x * 32 + j;
    double part[2] = {0, 0}, tot[2];
    // SU rows in flight per thread (branch-free clamped loads, consumed in row order: the sums are bit-identical to the one-row loop).
    // One row per iteration was one L2 round trip per row -- 32 of them in a row at 256 rows: 15 us per launch for 512 KB (round 6).
    constexpr int SU = 8;
    for (int r0 = ig; r0 < R; r0 += 8 * SU) {
        float v[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int r = r0 + 8 * u;
            v[u] = z[(size_t)(r < R ? r : ig) * C + c];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            if (r0 + 8 * u < R) {
                const double d = (double)v[u];
                part[0] += d;
                part[1] += d * d;
            }
        }
    }
    small_col_totals(sh, 2, part, ig, j, tot);
    const double m = tot[0] / (double)R;
    double var = tot[1] / (double)R - m * m;
    if (var < 0) var = 0;
    const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
    if (ig == 0) {
        mean[c] = mu;
        invstd[c] = is;
        if (running_mean) {
            const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
    }
    const float g = gamma[c], b = beta[c];
    for (int r0 = ig; r0 < R; r0 += 8 * SU) {
        float v[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int r = r0 + 8 * u;
            v[u] = z[(size_t)(r < R ? r : ig) * C + c];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int r = r0 + 8 * u;
            if (r < R) {
                float o = (v[u] - mu) * is * g + b;
                if (relu) o = fmaxf(o, 0.f);
                y[(size_t)r * C + c] = o;
            }
        }
    }
}

// g = dy (under the ReLU mask recomputed from z when zmask_beta is given); dbeta = sum g, dgamma = sum g xhat,
// dz = gamma invstd (g - dbeta / R - xhat dgamma / R); optionally dbias = sum dz (the Linear bias in front of the BatchNorm).
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ zmask_beta,
                                                            float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                            float* __restrict__ dbias, float* __restrict__ dz, int R, int C) {
    __shared__ double sh[2][8][32];
    const int j = threadIdx.x & 31, ig = threadIdx.x >> 5, c = blockIdx.x * 32 + j;
    const float mu = mean[c], is = invstd[c], ga = gamma[c];
    const float zb = zmask_beta ? zmask_beta[c] : 0.f;
    double part[2] = {0, 0}, tot[2];
    constexpr int SU = 8;                     // rows in flight per thread, consumed in row order (see bn_small_fwd_kernel)
    for (int r0 = ig; r0 < R; r0 += 8 * SU) {
        float zv[SU], gv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int r = r0 + 8 * u;
            const size_t o = (size_t)(r < R ? r : ig) * C + c;
            zv[u] = z[o];
            gv[u] = dy[o];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            if (r0 + 8 * u < R) {
                const float zz = zv[u];
                float g = gv[u];
                if (zmask_beta) g = (zz - mu) * is * ga + zb > 0.f ? g : 0.f;
                part[0] += (double)g;
                part[1] += (double)g * (double)((zz - mu) * is);
            }
        }
    }
    small_col_totals(sh, 2, part, ig, j, tot);
    const float db = (float)tot[0], dg = (float)tot[1];
    if (ig == 0) {
        if (dbeta) dbeta[c] = db;
        if (dgamma) dgamma[c] = dg;
    }
    const float invR = 1.f / (float)R;
    double sdz[1] = {0};
    for (int r0 = ig; r0 < R; r0 += 8 * SU) {
        float zv[SU], gv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int r = r0 + 8 * u;
            const size_t o = (size_t)(r < R ? r : ig) * C + c;
            zv[u] = z[o];
            gv[u] = dy[o];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int r = r0 + 8 * u;
            if (r < R) {
                const float zz = zv[u];
                float g = gv[u];
                if (zmask_beta) g = (zz - mu) * is * ga + zb > 0.f ? g : 0.f;
                const float xh = (zz - mu) * is;
                const float o = ga * is * (g - db * invR - xh * dg * invR);
                dz[(size_t)r * C + c] = o;
                sdz[0] += (double)o;
            }
        }
    }
    if (dbias) {
        double t[1];
        small_col_totals(sh, 1, sdz, ig, j, t);
        if (ig == 0) dbias[c] = (float)t[0];
    }
}

int bn_small_rows() {
    static const int v = getenv("SSAD_BN_SMALL") ? atoi(getenv("SSAD_BN_SMALL")) : 512;
    return v;
}

}  // namespace

// 1 when the one-launch BatchNorm kernels take R rows x C channels (SSAD_BN_SMALL=0 switches them off).
extern "C" int ssad_bn_small_ok(int64_t R, int C) { return R > 0 && R <= bn_small_rows() && C > 0 && C % 32 == 0; }

extern "C" int ssad_bn_small_fwd(const float* z, const float* gamma, const float* beta, float* y, float* mean, float* invstd,
                                 float* running_mean, float* running_var, int64_t R, int C, float eps, float momentum, int relu,
                                 void* stream) {
    SSAD_CHECK_ARG(z && gamma && beta && y && mean && invstd, "null pointer");
    SSAD_CHECK_ARG(ssad_bn_small_ok(R, C), "outside the small-batch range (ssad_bn_small_ok)");
    SSAD_CHECK_ARG(!running_mean == !running_var, "running_mean and running_var come together");
    hipLaunchKernelGGL(bn_small_fwd_kernel, dim3(C / 32), dim3(256), 0, (hipStream_t)stream, z, gamma, beta, y, mean, invstd,
                       running_mean, running_var, (int)R, C, eps, momentum, relu);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_bn_small_bwd(const float* dy, const float* z, const float* mean, const float* invstd, const float* gamma,
                                 const float* zmask_beta, float* dbeta, float* dgamma, float* dbias, float* dz, int64_t R, int C,
                                 void* stream) {
    SSAD_CHECK_ARG(dy && z && mean && invstd && gamma && dz, "null pointer");
    SSAD_CHECK_ARG(ssad_bn_small_ok(R, C), "outside the small-batch range (ssad_bn_small_ok)");
    hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(C / 32), dim3(256), 0, (hipStream_t)stream, dy, z, mean, invstd, gamma, zmask_beta,
                       dbeta, dgamma, dbias, dz, (int)R, C);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int64_t N, int H, int W, int C, int64_t dy_elems,
                                     void* stream) {
    SSAD_CHECK_ARG(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "bad argument");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    SSAD_CHECK_ARG(dy_elems == N * Ho * Wo * C, "dy does not hold N x Ho x Wo x C elements for this x");
    SSAD_CHECK_ARG(C % 4 == 0, "C must be a multiple of 4");
    const int64_t total = N * H * W * (C / 4);
    SSAD_CHECK_ARG(cdiv64(total, 256) < (int64_t)2147483647, "too large");
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, total,
                       H, W, C / 4, Ho, Wo);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_maxpool3x3s2_bwd_idx(const uint8_t* idx, const float* dy, float* dx, int64_t N, int H, int W, int C,
                                        int64_t dy_elems, void* stream) {
    SSAD_CHECK_ARG(idx && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bad argument");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    SSAD_CHECK_ARG(dy_elems == N * Ho * Wo * C, "dy / idx do not hold N x Ho x Wo x C elements for this dx");
    const int64_t total = N * H * W * (C / 4);
    SSAD_CHECK_ARG(cdiv64(total, 256) < (int64_t)2147483647, "too large");
    hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, idx, dy, dx,
                       total, H, W, C / 4, Ho, Wo);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// Stem head of the backward pass: dz (gradient of the raw conv1 output) and the BatchNorm parameter gradients from the
// POOLED gradient, the argmax slots and z -- max-pool backward, the ReLU mask and both BatchNorm passes without ever
// storing the gradient of the 128x128 activation.  workspace: ssad_colreduce_workspace(N*H*W, C) doubles.
template <typename T>
static int pool_bn_relu_bwd_impl(const uint8_t* idx, const T* dpool, const T* z, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, float* dbeta, float* dgamma,
                                 T* dz, int64_t N, int H, int W, int C, int64_t dpool_elems, double* workspace, void* stream) {
    SSAD_CHECK_ARG(idx && dpool && z && mean && invstd && gamma && beta && dbeta && dgamma && dz && workspace, "null pointer");
    constexpr int E = Lane<T>::E;
    SSAD_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % E == 0 && C <= 1024 && 256 % (C / E) == 0, "bad shape");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    SSAD_CHECK_ARG(dpool_elems == N * Ho * Wo * C, "dpool / idx do not hold N x Ho x Wo x C elements for this z");
    const int64_t R = N * H * W;
    const int RL = 256 / (C / E);
    const int64_t NB = N * ((H + 1) / 2) * ((W + 1) / 2);          // 2 x 2 pixel blocks: the unit a thread works on
    int64_t nblk = cdiv64(R, (int64_t)RL * 32);                      // = the rows ssad_colreduce_workspace(R, C) provides
    if (nblk > 2048) nblk = 2048;
    const int64_t rows_per_block = cdiv64(NB, nblk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((pool_bn_bwd_kernel<0, T>), dim3((unsigned)nblk), dim3(256), 0, st, idx, dpool, z, mean, invstd, gamma, beta,
                       (const float*)nullptr, (const float*)nullptr, (T*)nullptr, workspace, R, H, W, C, Ho, Wo, rows_per_block);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, workspace, (int)nblk, C, dbeta, dgamma);
    hipLaunchKernelGGL((pool_bn_bwd_kernel<1, T>), dim3((unsigned)nblk), dim3(256), 0, st, idx, dpool, z, mean, invstd, gamma, beta,
                       dbeta, dgamma, dz, (double*)nullptr, R, H, W, C, Ho, Wo, rows_per_block);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// The apply pass alone: dbeta / dgamma are INPUTS (taken over the pooled tensors: ssad_bn_relu_maxpool_fwd_win, ssad_bn_bwd_reduce_zmask).
template <typename T>
static int pool_bn_relu_bwd_apply_impl(const uint8_t* idx, const T* dpool, const T* z, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, const float* dbeta, const float* dgamma, T* dz, int64_t N,
                                       int H, int W, int C, int64_t dpool_elems, void* stream) {
    constexpr int E = Lane<T>::E;
    SSAD_CHECK_ARG(idx && dpool && z && mean && invstd && gamma && beta && dbeta && dgamma && dz, "null pointer");
    SSAD_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % E == 0 && C <= 1024 && 256 % (C / E) == 0, "bad shape");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    SSAD_CHECK_ARG(dpool_elems == N * Ho * Wo * C, "dpool / idx do not hold N x Ho x Wo x C elements for this z");
    const int64_t R = N * H * W;
    const int RL = 256 / (C / E);
    const int64_t NB = N * ((H + 1) / 2) * ((W + 1) / 2);
    int64_t nblk = cdiv64(R, (int64_t)RL * 32);
    if (nblk > 2048) nblk = 2048;
    const int64_t rows_per_block = cdiv64(NB, nblk);
    hipLaunchKernelGGL((pool_bn_bwd_kernel<1, T>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, idx, dpool, z, mean, invstd, gamma,
                       beta, dbeta, dgamma, dz, (double*)nullptr, R, H, W, C, Ho, Wo, rows_per_block);
    SSAD_CHECK_LAUNCH();
    return 0;
}
extern "C" int ssad_pool_bn_relu_bwd_apply(const uint8_t* idx, const float* dpool, const float* z, const float* mean, const float* invstd,
                                           const float* gamma, const float* beta, const float* dbeta, const float* dgamma, float* dz,
                                           int64_t N, int H, int W, int C, int64_t dpool_elems, void* stream) {
    return pool_bn_relu_bwd_apply_impl<float>(idx, dpool, z, mean, invstd, gamma, beta, dbeta, dgamma, dz, N, H, W, C, dpool_elems, stream);
}
extern "C" int ssad_pool_bn_relu_bwd_apply_h(const uint8_t* idx, const void* dpool, const void* z, const float* mean, const float* invstd,
                                             const float* gamma, const float* beta, const float* dbeta, const float* dgamma, void* dz,
                                             int64_t N, int H, int W, int C, int64_t dpool_elems, void* stream) {
    return pool_bn_relu_bwd_apply_impl<hf>(idx, (const hf*)dpool, (const hf*)z, mean, invstd, gamma, beta, dbeta, dgamma, (hf*)dz, N, H, W,
                                           C, dpool_elems, stream);
}

extern "C" int ssad_pool_bn_relu_bwd(const uint8_t* idx, const float* dpool, const float* z, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, float* dbeta, float* dgamma,
                                     float* dz, int64_t N, int H, int W, int C, int64_t dpool_elems, double* workspace, void* stream) {
    return pool_bn_relu_bwd_impl<float>(idx, dpool, z, mean, invstd, gamma, beta, dbeta, dgamma, dz, N, H, W, C, dpool_elems, workspace, stream);
}

extern "C" int ssad_pool_bn_relu_bwd_h(const uint8_t* idx, const void* dpool, const void* z, const float* mean,
                                       const float* invstd, const float* gamma, const float* beta, float* dbeta, float* dgamma,
                                       void* dz, int64_t N, int H, int W, int C, int64_t dpool_elems, double* workspace, void* stream) {
    return pool_bn_relu_bwd_impl<hf>(idx, (const hf*)dpool, (const hf*)z, mean, invstd, gamma, beta, dbeta, dgamma, (hf*)dz, N, H, W, C,
                                     dpool_elems, workspace, stream);
}

extern "C" int ssad_gap_bwd(const float* dpooled, float* dy, int64_t N, int HW, int C, int stride, int offset, int accumulate,
                            void* stream) {
    SSAD_CHECK_ARG(dpooled && dy && N > 0 && HW > 0 && C > 0 && offset >= 0 && offset + C <= stride, "bad argument");
    const int64_t total = N * HW * C;
    if (C % 4 == 0 && ((uintptr_t)dy & 15) == 0)
        hipLaunchKernelGGL(gap_bwd4_kernel<float>, dim3((unsigned)cdiv64(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, dpooled, dy,
                           total / 4, HW, C / 4, stride, offset, accumulate);
    else
        hipLaunchKernelGGL(gap_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, dpooled, dy, total,
                           HW, C, stride, offset, accumulate);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_gap_bwd_h(const float* dpooled, void* dy, int64_t N, int HW, int C, int stride, int offset, int accumulate,
                              void* stream) {
    SSAD_CHECK_ARG(dpooled && dy && N > 0 && HW > 0 && C > 0 && C % 4 == 0 && offset >= 0 && offset + C <= stride, "bad argument");
    const int64_t total = N * HW * C;
    hipLaunchKernelGGL(gap_bwd4_kernel<hf>, dim3((unsigned)cdiv64(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, dpooled, (hf*)dy,
                       total / 4, HW, C / 4, stride, offset, accumulate);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_stem_im2col(const float* img, float* col, int64_t B, int H, int W, int Hv, int Wv, void* stream) {
    SSAD_CHECK_ARG(img && col && B > 0 && H > 0 && W > 0 && Hv > 0 && Wv > 0, "bad argument");
    const int Ho = (Hv - 1) / 2 + 1, Wo = (Wv - 1) / 2 + 1;
    const int64_t total4 = B * Ho * Wo * 40;
    SSAD_CHECK_ARG(cdiv64(total4, 256) < (int64_t)2147483647, "too large");
    hipLaunchKernelGGL(stem_im2col_kernel, dim3((unsigned)cdiv64(total4, 256)), dim3(256), 0, (hipStream_t)stream, img, col, total4,
                       H, W, Hv, Wv, Ho, Wo);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_pack_stem_weight_2d(const float* w_oihw, float* out, void* stream) {
    SSAD_CHECK_ARG(w_oihw && out, "null pointer");
    hipLaunchKernelGGL(pack_stem_weight_2d_kernel, dim3((64 * 160 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, out);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_flip_transpose_weight(const float* w_ohwi, float* out, int O, int I, int KH, int KW, void* stream) {
    SSAD_CHECK_ARG(w_ohwi && out && O > 0 && I > 0 && KH > 0 && KW > 0, "bad argument");
    const int64_t total = (int64_t)O * I * KH * KW;
    hipLaunchKernelGGL(flip_transpose_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi, out, O,
                       I, KH, KW);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_softmax_ce(const float* logits, const int64_t* labels, int B, int C, float* loss_acc, float* dlogits,
                               int ldd, float grad_scale, void* stream) {
    SSAD_CHECK_ARG(logits && labels && loss_acc && B > 0 && C > 0, "bad argument");
    SSAD_CHECK_ARG(!dlogits || ldd >= C, "dlogits rows shorter than the class count");
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, B, C, loss_acc, dlogits, ldd,
                       grad_scale);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_sgd_step(float* p, const float* g, float* m, int64_t n, float lr, float momentum, float weight_decay,
                             float grad_scale, void* stream) {
    SSAD_CHECK_ARG(p && g && m && n > 0, "bad argument");
    hipLaunchKernelGGL(sgd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, n, lr, momentum, weight_decay,
                       grad_scale);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// SGD with the hyper-parameters read from device memory: hyper = [lr, momentum, weight_decay, grad_scale] (4 floats);
// scaler = NULL or the GradScaler state [loss_scale, growth_tracker, found_inf] (the update then uses grad_scale /
// loss_scale and is skipped while found_inf is set).  Same arithmetic as ssad_sgd_step.
extern "C" int ssad_sgd_step_dev(float* p, const float* g, float* m, int64_t n, const float* hyper, const float* scaler,
                                 void* stream) {
    SSAD_CHECK_ARG(p && g && m && hyper && n > 0, "bad argument");
    hipLaunchKernelGGL(sgd_dev_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, n, hyper, scaler);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// GradScaler pieces (see above): x *= scaler[0];  scaler[2] = 1 if any of g[0..n) is inf / nan;  the end-of-step update.
extern "C" int ssad_scale_by_loss_scale(float* x, int64_t n, const float* scaler, void* stream) {
    SSAD_CHECK_ARG(x && scaler && n > 0, "bad argument");
    hipLaunchKernelGGL(scale_by_dev_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, scaler);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_check_finite(const float* g, int64_t n, float* scaler, void* stream) {
    SSAD_CHECK_ARG(g && scaler && n > 0, "bad argument");
    hipLaunchKernelGGL(check_finite_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, g, n, scaler);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_loss_scaler_update(float* scaler, float growth_factor, float backoff_factor, int growth_interval,
                                       void* stream) {
    SSAD_CHECK_ARG(scaler && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval > 0,
                   "bad argument");
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler, growth_factor, backoff_factor,
                       (float)growth_interval);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// BatchNorm backward with g = dy * mask (mask4 from ssad_bn_apply_fwd_mask; NULL = no ReLU, g = dy): the two reductions,
// then dz = gamma * invstd * (g - dbeta / R - xhat * dgamma / R).  The gradient of the identity branch is NOT written: it is
// dy under the same mask, which the consumer applies itself (ssad_conv_igemm_dgrad_masked / ssad_conv3x3_c64).
extern "C" int ssad_bn_bwd_reduce_mask(const float* dy, const uint8_t* mask4, const float* z, const float* mean,
                                       const float* invstd, float* dbeta, float* dgamma, int64_t R, int C, double* workspace,
                                       void* stream) {
    SSAD_CHECK_ARG(z, "z required");
    return bn_bwd_reduce_impl<float>(dy, nullptr, z, mean, invstd, dbeta, dgamma, R, C, workspace, stream, nullptr, nullptr, mask4);
}

extern "C" int ssad_bn_apply_bwd_mask(const float* dy, const uint8_t* mask4, const float* z, const float* mean, const float* invstd,
                                      const float* gamma, const float* dbeta, const float* dgamma, float* dz, int64_t R, int C,
                                      void* stream) {
    return bn_apply_bwd_impl<float>(dy, nullptr, z, mean, invstd, gamma, dbeta, dgamma, dz, nullptr, R, C, 0, stream, nullptr, mask4);
}

extern "C" int ssad_bn_bwd_reduce_mask_h(const void* dy, const uint8_t* mask4, const void* z, const float* mean,
                                         const float* invstd, float* dbeta, float* dgamma, int64_t R, int C, double* workspace,
                                         void* stream) {
    SSAD_CHECK_ARG(z, "z required");
    return bn_bwd_reduce_impl<hf>((const hf*)dy, nullptr, (const hf*)z, mean, invstd, dbeta, dgamma, R, C, workspace, stream, nullptr, nullptr, mask4);
}

extern "C" int ssad_bn_apply_bwd_mask_h(const void* dy, const uint8_t* mask4, const void* z, const float* mean, const float* invstd,
                                        const float* gamma, const float* dbeta, const float* dgamma, void* dz, int64_t R, int C,
                                        void* stream) {
    return bn_apply_bwd_impl<hf>((const hf*)dy, nullptr, (const hf*)z, mean, invstd, gamma, dbeta, dgamma, (hf*)dz, nullptr, R, C, 0, stream, nullptr,
                                 mask4);
}

// ssad_flip_transpose_weight for n filters at once: desc[k] = {src offset, dst offset, O, I, KH, KW} (floats, host
// memory), sources inside `src`, results inside `dst`.  One launch per 32 filters (the kernel's table travels as a launch
// argument): PeraNet() with its default head is one launch, a deeper latent_space_layers two or more.
// fp32 -> half copy of a parameter arena (8 elements per thread; n % 8 == 0 or the tail is done element-wise)
__global__ void cvt_f32_f16_kernel(const float* __restrict__ src, hf* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const f32x4 a = *(const f32x4*)(src + i), b = *(const f32x4*)(src + i + 4);
        *(f16x8*)(dst + i) = f16x8{(hf)a[0], (hf)a[1], (hf)a[2], (hf)a[3], (hf)b[0], (hf)b[1], (hf)b[2], (hf)b[3]};
    } else {
        for (int64_t k = i; k < n; ++k) dst[k] = (hf)src[k];
    }
}

// The precision-16 step with half tensors reads its weights as halves: one rounded copy of the (fp32 master) arena per step, as
// torch.autocast makes one per use (cast cache) under pl.Trainer(precision=16), tools.py:263.  src, dst 16-byte aligned.
extern "C" int ssad_cvt_f32_f16(const float* src, void* dst, int64_t n, void* stream) {
    SSAD_CHECK_ARG(src && dst && n > 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "bad argument");
    hipLaunchKernelGGL(cvt_f32_f16_kernel, dim3((unsigned)cdiv64(cdiv64(n, 8), 256)), dim3(256), 0, (hipStream_t)stream, src, (hf*)dst, n);
    SSAD_CHECK_LAUNCH();
    return 0;
}

static int flip_transpose_batch_impl(const float* src, void* dst, int dst_half, const int64_t* desc, int n, void* stream) {
    SSAD_CHECK_ARG(src && dst && desc && n > 0, "bad argument");
    for (int k = 0; k < n; ++k)
        SSAD_CHECK_ARG(desc[6 * k + 2] > 0 && desc[6 * k + 3] > 0 && desc[6 * k + 4] > 0 && desc[6 * k + 5] > 0, "bad filter shape");
    for (int base = 0; base < n; base += 32) {
        FlipTable t;
        t.n = n - base < 32 ? n - base : 32;
        int64_t acc = 0;
        for (int k = 0; k < t.n; ++k) {
            for (int j = 0; j < 6; ++j) t.e[k][j] = desc[6 * (base + k) + j];
            t.e[k][6] = acc;
            acc += t.e[k][4] * t.e[k][5] * ((t.e[k][2] + 31) / 32) * ((t.e[k][3] + 31) / 32);     // taps x (o, i) tiles
        }
        t.total = acc;
        SSAD_CHECK_ARG(acc < (int64_t)2147483647, "too many tiles");
        if (dst_half) hipLaunchKernelGGL(flip_transpose_batch_kernel<hf>, dim3((unsigned)acc), dim3(256), 0, (hipStream_t)stream, src, (hf*)dst, t);
        else hipLaunchKernelGGL(flip_transpose_batch_kernel<float>, dim3((unsigned)acc), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, t);
        SSAD_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int ssad_flip_transpose_batch(const float* src, float* dst, const int64_t* desc, int n, void* stream) {
    return flip_transpose_batch_impl(src, dst, 0, desc, n, stream);
}

// the flipped filters written as halves (dgrad operands of the precision-16 step with half tensors)
extern "C" int ssad_flip_transpose_batch_h(const float* src, void* dst, const int64_t* desc, int n, void* stream) {
    return flip_transpose_batch_impl(src, dst, 1, desc, n, stream);
}
