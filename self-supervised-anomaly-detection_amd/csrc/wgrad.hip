// Weight gradient of conv / linear layers on the fp32 matrix cores, split over pixel ranges.
//
//   dW[co][ky][kx][ci] = sum_m dY[m][co] * X[n(m)][oy(m)*s + ky - p][ox(m)*s + kx - p][ci]
//
// Replaces the wgrad half of autograd's conv2d / linear backward that the reference gets from
// loss.backward() inside pl.Trainer.fit (src/self_supervised/tools.py:270, :303; models.py:256-277).
//
// MFMA formulation: D[co][ci] += A[co][m] * B[m][ci] with the contraction over output pixels m.  Both operands
// are read from NHWC rows as they lie in HBM (dY row = Cout contiguous floats, X row = Cin contiguous floats),
// staged [32 pixels][BT channels] in LDS; lane (r, h) of v_mfma_f32_32x32x2_f32 reads pixel 2*kk+h, channel r:
// 32 consecutive floats per lane half, conflict-free ds_read_b32.  A workgroup owns one (tap, co-tile, ci-tile)
// and one pixel range; partial tiles go to slab[split] and ssad_wgrad_reduce sums the splits in a fixed order
// (deterministic, no float atomics), writing OIHW (checkpoint layout) or OHWI.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int PK = 32;   // pixels per staging step

__device__ __attribute__((aligned(16))) float g_wzero[8] = {0, 0, 0, 0, 0, 0, 0, 0};

struct WgradParams {
    const float* dy;
    const float* x;
    float* slab;
    int64_t M, chunk;
    int H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int co_tiles, ci_tiles, splits;
};

// Wave-specialised: 8 waves, waves 0-3 ("consumers", one per SIMD) issue nothing but LDS fragment reads and MFMAs,
// waves 4-7 ("loaders", the second wave of each SIMD) do the address arithmetic, the global loads and the LDS stores of
// the NEXT steps.  Measured on the first, 4-wave form of this kernel (every wave loads and computes): each
// global_load_dwordx4 costs the issuing wave ~60-70 cycles of its in-order stream -- 8 per 64 MFMAs = 14 % of the
// step, wherever in the step they are placed -- and here those cycles are spent by a wave that has no MFMAs to delay
// (layer2-4 wgrad 0.74 -> 0.69 ms).  One block-wide barrier per step hands buffer
// (s+1)&1 from the loaders to the consumers; loads for step s+2 are in flight during step s+1's hand-over.
template <int BT>
__global__ __launch_bounds__(512, 4) void wgrad_f32_kernel(WgradParams p) {
    constexpr int T = BT / 64;            // 32x32 tiles per consumer wave per dim (consumer waves 2x2)
    constexpr int F4 = BT / 4;            // float4 per staged row
    constexpr int RPP = 256 / F4;         // rows per pass (256 loader threads)
    constexpr int NP = PK / RPP;          // passes
    constexpr int STAGE = 2 * PK * BT;    // floats per stage (dY tile + X tile)
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= 4;
    const int r = lane & 31, h = lane >> 5;
    const int wm = (wave & 3) >> 1, wn = wave & 1;
    const int ntiles = p.KH * p.KW * p.co_tiles * p.ci_tiles;
    const int64_t jj = blockIdx.x >> 3;
    const int split = (int)(jj / ntiles) * 8 + (int)(blockIdx.x & 7);
    if (split >= p.splits) return;
    int tile = (int)(jj % ntiles);
    const int ci_t = tile % p.ci_tiles; tile /= p.ci_tiles;
    const int co_t = tile % p.co_tiles;
    const int tap = tile / p.co_tiles;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int co0 = co_t * BT, ci0 = ci_t * BT;
    const int64_t m_begin = (int64_t)split * p.chunk;
    const int64_t m_end = m_begin + p.chunk < p.M ? m_begin + p.chunk : p.M;
    const int len = (int)(m_end - m_begin);
    const int nsteps = (len + PK - 1) / PK;
    const int HoWo = p.Ho * p.Wo;

    if (loader) {
        const int lt = tid - 256;
        const int c4 = lt % F4, r0 = lt / F4;
        const bool co_ok = co0 + c4 * 4 < p.Cout, ci_ok = ci0 + c4 * 4 < p.Cin;
        const float* zero = g_wzero;
        const int dn = PK / HoWo, rp = PK - dn * HoWo;
        const int dyy = rp / p.Wo, dxx = rp - dyy * p.Wo;
        const int64_t d_main = (((int64_t)dn * p.H + dyy * p.stride) * p.W + dxx * p.stride) * p.Cin;
        const int64_t d_wx = ((int64_t)p.stride * p.W - (int64_t)p.Wo * p.stride) * p.Cin;
        const int64_t d_wy = ((int64_t)p.H - (int64_t)p.Ho * p.stride) * p.W * p.Cin;
        int roy[NP], rox[NP];
        const float* yptr[NP];
        const float* xptr[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int64_t m = m_begin + r0 + i * RPP;
            const int64_t n = m / HoWo;
            const int rem = (int)(m - n * HoWo);
            roy[i] = rem / p.Wo;
            rox[i] = rem - roy[i] * p.Wo;
            yptr[i] = p.dy + m * p.Cout + co0 + c4 * 4;
            xptr[i] = p.x + ci0 + c4 * 4 +
                      ((n * p.H + (roy[i] * p.stride - p.pad + ky)) * p.W + (rox[i] * p.stride - p.pad + kx)) * p.Cin;
        }
        const int64_t ystep = (int64_t)PK * p.Cout;
        int lrow = r0;
        f32x4 ry[NP], rx[NP];
        auto load_step = [&]() {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const bool live = lrow + i * RPP < len;
                const int iy = roy[i] * p.stride - p.pad + ky, ix = rox[i] * p.stride - p.pad + kx;
                const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                ry[i] = *(const f32x4*)((live && co_ok) ? yptr[i] : zero);
                rx[i] = *(const f32x4*)((live && inb && ci_ok) ? xptr[i] : zero);
                yptr[i] += ystep;
                xptr[i] += d_main;
                int ox = rox[i] + dxx, oy = roy[i] + dyy;
                if (ox >= p.Wo) { ox -= p.Wo; ++oy; xptr[i] += d_wx; }
                if (oy >= p.Ho) { oy -= p.Ho; xptr[i] += d_wy; }
                rox[i] = ox; roy[i] = oy;
            }
            lrow += PK;
        };
        auto store_step = [&](float* buf) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                *(f32x4*)(buf + (r0 + i * RPP) * BT + c4 * 4) = ry[i];
                *(f32x4*)(buf + PK * BT + (r0 + i * RPP) * BT + c4 * 4) = rx[i];
            }
        };
        if (nsteps > 0) {
            load_step();
            store_step(lds);
            if (nsteps > 1) load_step();                      // step 1 in flight
        }
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) store_step(lds + ((s + 1) & 1) * STAGE);
            if (s + 2 < nsteps) load_step();
            __syncthreads();
        }
        return;
    }

    // ---- consumers ----
    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const float* cur = lds + (s & 1) * STAGE;
        const float* ya = cur + h * BT + wm * 32 * T + r;
        const float* xb = cur + PK * BT + h * BT + wn * 32 * T + r;
        float a[2][T], b[2][T];
#pragma unroll
        for (int i = 0; i < T; ++i) a[0][i] = ya[i * 32];
#pragma unroll
        for (int j = 0; j < T; ++j) b[0][j] = xb[j * 32];
#pragma unroll
        for (int kk = 0; kk < PK / 2; ++kk) {
            const int cu = kk & 1, nx = cu ^ 1;
            if (kk + 1 < PK / 2) {
#pragma unroll
                for (int i = 0; i < T; ++i) a[nx][i] = ya[(kk + 1) * 2 * BT + i * 32];
#pragma unroll
                for (int j = 0; j < T; ++j) b[nx][j] = xb[(kk + 1) * 2 * BT + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the reads one pixel pair ahead of the MFMAs (see above)
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j) acc[i][j] = mfma32(a[cu][i], b[cu][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    const int taps = p.KH * p.KW;
    float* out = p.slab + (int64_t)split * p.Cout * taps * p.Cin;
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const int ci = ci0 + (wn * T + j) * 32 + r;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + (wm * T + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < p.Cout) out[((int64_t)co * taps + tap) * p.Cin + ci] = acc[i][j][e];
            }
    }
}

template <bool F16> struct OpType;
template <> struct OpType<false> {
    using t = __bf16; using v4 = bf16x4; using v8 = bf16x8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c, int, int, int) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct OpType<true> {
    using t = _Float16; using v4 = f16x4; using v8 = f16x8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c, int, int, int) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// bf16-operand variant (Trainer(precision=16)): same contraction, operands rounded to bf16 while staging.
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive k (= pixels) per lane for a fixed channel, so the tiles are stored
// TRANSPOSED in LDS, [channel][32 pixels + pad] bf16 (80-byte rows): each thread loads a 4-pixel x 4-channel block
// (four 16-byte global loads), converts, and writes four 8-byte column pieces (lanes run over pixel groups first:
// conflict-free).  fp32 accumulate, fp32 slabs, same deterministic reduce.
// NS: number of bf16 parts per operand, staged side by side in the LDS row (see conv_igemm.hip, BF = 3 / 6):
//     1  bf16 operands;  2  hi, lo and three MFMAs per tile (lo*hi + hi*lo + hi*hi: ~2^-17 per product, "bf16x3");
//     3  hi, mid, lo and the six products of weight >= 2^-18 (fp32-faithful, "bf16x6"; one LDS stage so that two
//        workgroups still fit a CU).
// WS: wave-specialised like wgrad_f32_kernel (4 MFMA waves + 4 loader waves): with MFMAs 16x shorter than the fp32 ones
// the loads, converts and transposed LDS stores dominate a 4-wave step.
// F16: fp16 operands (v_mfma_f32_32x32x16_f16) instead of bf16 -- the reference's fp16 autocast (tools.py:263); NS = 1 only.
// TI = hf: dy and x are stored as halves (the precision-16 step with half tensors): 8-byte loads, no conversion of value
template <int BT, int NS = 1, bool WS = false, bool F16 = false, typename TI = float>
__global__ __launch_bounds__(WS ? 512 : 256, WS ? 4 : 2) void wgrad_bf16_kernel(WgradParams p) {
    using op_t = typename OpType<F16>::t;
    using op4 = typename OpType<F16>::v4;
    using op8 = typename OpType<F16>::v8;
    static_assert(!F16 || NS == 1, "fp16 operands: plain products only");
    constexpr bool X3 = NS >= 2;
    constexpr bool DBUF = NS < 3;                // bf16x6: single stage (53 KB at BT = 128)
    static_assert(!WS || DBUF, "wave specialisation needs the double-buffered stages");
    constexpr int T = BT / 64;
    constexpr int LDP = NS * PK + 8;             // bf16 elements per LDS row ([32 hi | 32 mid | 32 lo] + pad)
    constexpr int STAGE = DBUF ? 2 * BT * LDP : 0;      // bf16 elements between the two stages (dY^T tile + X^T tile each)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    op_t* L = (op_t*)lds;

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
    const bool loader = WS && tid >= 256;
    const int lt = tid & 255;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.KH * p.KW * p.co_tiles * p.ci_tiles;
    const int64_t jj = blockIdx.x >> 3;
    const int split = (int)(jj / ntiles) * 8 + (int)(blockIdx.x & 7);
    if (split >= p.splits) return;
    int tile = (int)(jj % ntiles);
    const int ci_t = tile % p.ci_tiles; tile /= p.ci_tiles;
    const int co_t = tile % p.co_tiles;
    const int tap = tile / p.co_tiles;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int co0 = co_t * BT, ci0 = ci_t * BT;
    const int64_t m_begin = (int64_t)split * p.chunk;
    const int64_t m_end = m_begin + p.chunk < p.M ? m_begin + p.chunk : p.M;
    const int HoWo = p.Ho * p.Wo;
    const TI* zero = (const TI*)g_wzero;

    const int pg = lt & 7, c4 = lt >> 3;        // pixel group (4 pixels), channel quad
    const bool stager = c4 < BT / 4 && (!WS || loader);
    const bool co_ok = stager && co0 + c4 * 4 < p.Cout, ci_ok = stager && ci0 + c4 * 4 < p.Cin;

    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // coordinates of this thread's first pixel, advanced by PK per step
    int64_t m1 = m_begin + pg * 4;
    int rn, roy, rox;
    {
        const int64_t n = m1 / HoWo;
        const int rem = (int)(m1 - n * HoWo);
        rn = (int)n; roy = rem / p.Wo; rox = rem - roy * p.Wo;
    }
    const int dn = PK / HoWo, rp = PK - dn * HoWo;
    const int dyy = rp / p.Wo, dxx = rp - dyy * p.Wo;
    const TI* ybase = (const TI*)p.dy + co0 + c4 * 4;
    const TI* xbase = (const TI*)p.x + ci0 + c4 * 4;

    typename Raw4<TI>::t ry[4], rx[4];          // fp32 tensors: floats, rounded in store_step; half tensors: the stored halves
    auto load_step = [&]() {
        int n = rn, oy = roy, ox = rox;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool live = m1 + q < m_end;
            const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
            const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            ry[q] = ldraw4((live && co_ok) ? ybase + (m1 + q) * p.Cout : zero);
            rx[q] = ldraw4((live && inb && ci_ok) ? xbase + (((int64_t)n * p.H + iy) * p.W + ix) * p.Cin : zero);
            if (++ox >= p.Wo) { ox = 0; if (++oy >= p.Ho) { oy = 0; ++n; } }
        }
        m1 += PK;
        int ox2 = rox + dxx, oy2 = roy + dyy, n2 = rn + dn;
        if (ox2 >= p.Wo) { ox2 -= p.Wo; ++oy2; }
        if (oy2 >= p.Ho) { oy2 -= p.Ho; ++n2; }
        rox = ox2; roy = oy2; rn = n2;
    };
    auto store_step = [&](op_t* buf) {
        if (!stager) return;
        op_t* Yt = buf;
        op_t* Xt = buf + BT * LDP;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            op4 vy = {(op_t)ry[0][k], (op_t)ry[1][k], (op_t)ry[2][k], (op_t)ry[3][k]};
            op4 vx = {(op_t)rx[0][k], (op_t)rx[1][k], (op_t)rx[2][k], (op_t)rx[3][k]};
            *(op4*)(Yt + (c4 * 4 + k) * LDP + pg * 4) = vy;
            *(op4*)(Xt + (c4 * 4 + k) * LDP + pg * 4) = vx;
            if (X3) {
                op4 ly = {(op_t)(ry[0][k] - (float)vy[0]), (op_t)(ry[1][k] - (float)vy[1]),
                             (op_t)(ry[2][k] - (float)vy[2]), (op_t)(ry[3][k] - (float)vy[3])};
                op4 lx = {(op_t)(rx[0][k] - (float)vx[0]), (op_t)(rx[1][k] - (float)vx[1]),
                             (op_t)(rx[2][k] - (float)vx[2]), (op_t)(rx[3][k] - (float)vx[3])};
                *(op4*)(Yt + (c4 * 4 + k) * LDP + PK + pg * 4) = ly;
                *(op4*)(Xt + (c4 * 4 + k) * LDP + PK + pg * 4) = lx;
                if (NS == 3) {
                    op4 my, mx;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        my[q] = (op_t)(ry[q][k] - (float)vy[q] - (float)ly[q]);
                        mx[q] = (op_t)(rx[q][k] - (float)vx[q] - (float)lx[q]);
                    }
                    *(op4*)(Yt + (c4 * 4 + k) * LDP + 2 * PK + pg * 4) = my;
                    *(op4*)(Xt + (c4 * 4 + k) * LDP + 2 * PK + pg * 4) = mx;
                }
            }
        }
    };

    const int nsteps = (int)((m_end - m_begin + PK - 1) / PK);
    auto compute_step = [&](const op_t* cur) {
        const op_t* ya = cur + (wm * 32 * T + r) * LDP + h * 8;
        const op_t* xb = cur + BT * LDP + (wn * 32 * T + r) * LDP + h * 8;
#pragma unroll
        for (int k16 = 0; k16 < PK / 16; ++k16) {
            op8 a[T], b[T];
#pragma unroll
            for (int i = 0; i < T; ++i) a[i] = *(const op8*)(ya + i * 32 * LDP + k16 * 16);
#pragma unroll
            for (int j = 0; j < T; ++j) b[j] = *(const op8*)(xb + j * 32 * LDP + k16 * 16);
            if (NS == 3) {                      // parts: a / al / am = hi / mid / lo  (x = hi + mid + lo)
                op8 al[T], bl[T], am[T], bm[T];
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    al[i] = *(const op8*)(ya + i * 32 * LDP + PK + k16 * 16);
                    am[i] = *(const op8*)(ya + i * 32 * LDP + 2 * PK + k16 * 16);
                }
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    bl[j] = *(const op8*)(xb + j * 32 * LDP + PK + k16 * 16);
                    bm[j] = *(const op8*)(xb + j * 32 * LDP + 2 * PK + k16 * 16);
                }
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j) {          // smallest terms first; hi*hi is added by the common tail below
                        acc[i][j] = OpType<F16>::mfma(al[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = OpType<F16>::mfma(am[i], b[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = OpType<F16>::mfma(a[i], bm[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = OpType<F16>::mfma(al[i], b[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = OpType<F16>::mfma(a[i], bl[j], acc[i][j], 0, 0, 0);
                    }
            } else if (X3) {
                op8 al[T], bl[T];
#pragma unroll
                for (int i = 0; i < T; ++i) al[i] = *(const op8*)(ya + i * 32 * LDP + PK + k16 * 16);
#pragma unroll
                for (int j = 0; j < T; ++j) bl[j] = *(const op8*)(xb + j * 32 * LDP + PK + k16 * 16);
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j) {
                        acc[i][j] = OpType<F16>::mfma(al[i], b[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = OpType<F16>::mfma(a[i], bl[j], acc[i][j], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j) acc[i][j] = OpType<F16>::mfma(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    if (WS) {
        if (loader) {
            if (nsteps > 0) {
                load_step();
                store_step(L);
                if (nsteps > 1) load_step();
            }
            __syncthreads();
            for (int s = 0; s < nsteps; ++s) {
                if (s + 1 < nsteps) store_step(L + ((s + 1) & 1) * STAGE);
                if (s + 2 < nsteps) load_step();
                __syncthreads();
            }
            return;
        }
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            compute_step(L + (s & 1) * STAGE);
            __syncthreads();
        }
    } else {
    if (nsteps > 0) {
        load_step();
        store_step(L);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const op_t* cur = L + (DBUF ? (s & 1) * STAGE : 0);
        const bool more = s + 1 < nsteps;
        if (more) load_step();
        compute_step(cur);
        if (DBUF) {
            if (more) store_step(L + ((s + 1) & 1) * STAGE);
            __syncthreads();
        } else {
            __syncthreads();                    // every wave is done reading the stage
            if (more) store_step(L);
            __syncthreads();
        }
    }
    }

    const int taps = p.KH * p.KW;
    float* out = p.slab + (int64_t)split * p.Cout * taps * p.Cin;
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const int ci = ci0 + (wn * T + j) * 32 + r;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + (wm * T + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < p.Cout) out[((int64_t)co * taps + tap) * p.Cin + ci] = acc[i][j][e];
            }
    }
}

// out[...] = sum_s slab[s][co][k] for k < KH*KW*Cin_real; slab rows are Kpad floats long.
// 256 threads = 32 consecutive outputs x 8 split lanes: lane g adds splits g, g+8, ... in order, the 8 lane sums are
// added in a fixed order through LDS (deterministic), so a reduction over ~1000 splits is not one long serial chain.
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int splits, int Cout, int Kpad,
                                    int KH, int KW, int Cin, int to_oihw, int accumulate) {
    __shared__ float part[8][33];
    const int lx = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t idx = (int64_t)blockIdx.x * 32 + lx;
    const int Kreal = KH * KW * Cin;
    const bool ok = idx < (int64_t)Cout * Kreal;
    const int co = ok ? (int)(idx / Kreal) : 0, k = ok ? (int)(idx - (int64_t)co * Kreal) : 0;
    const int64_t stride = (int64_t)Cout * Kpad;
    const float* s = slab + (int64_t)co * Kpad + k;
    float v = 0.f;
    if (ok)
        for (int i = g; i < splits; i += 8) v += s[i * stride];
    part[g][lx] = v;
    __syncthreads();
    if (g == 0 && ok) {
        float t = part[0][lx];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += part[q][lx];
        int64_t o = idx;
        if (to_oihw) {
            const int tap = k / Cin, ci = k - tap * Cin;
            o = ((int64_t)co * Cin + ci) * (KH * KW) + tap;
        }
        out[o] = accumulate ? out[o] + t : t;
    }
}

// The same sums, four consecutive outputs per thread (16-byte loads) and four splits in flight: identical order of additions per
// output (lane g adds splits g, g + 8, ... in order, then the eight lane sums in order), so the result is bit-identical to the
// one-float form; needs Kreal % 4 == 0 and Kpad % 4 == 0.
__global__ void wgrad_reduce4_kernel(const float* __restrict__ slab, float* __restrict__ out, int splits, int Cout, int Kpad,
                                     int KH, int KW, int Cin, int to_oihw, int accumulate) {
    __shared__ f32x4 part[8][33];
    const int lx = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t idx = ((int64_t)blockIdx.x * 32 + lx) * 4;
    const int Kreal = KH * KW * Cin;
    const bool ok = idx < (int64_t)Cout * Kreal;
    const int co = ok ? (int)(idx / Kreal) : 0, k = ok ? (int)(idx - (int64_t)co * Kreal) : 0;
    const int64_t stride = (int64_t)Cout * Kpad;
    const float* s = slab + (int64_t)co * Kpad + k;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        int i = g;
        for (; i + 24 < splits; i += 32) {
            const f32x4 a = *(const f32x4*)(s + (int64_t)i * stride), b = *(const f32x4*)(s + (int64_t)(i + 8) * stride);
            const f32x4 c = *(const f32x4*)(s + (int64_t)(i + 16) * stride), d = *(const f32x4*)(s + (int64_t)(i + 24) * stride);
            v += a; v += b; v += c; v += d;
        }
        for (; i < splits; i += 8) v += *(const f32x4*)(s + (int64_t)i * stride);
    }
    part[g][lx] = v;
    __syncthreads();
    if (g == 0 && ok) {
        f32x4 t = part[0][lx];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += part[q][lx];
        if (to_oihw) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = k + j, tap = kk / Cin, ci = kk - tap * Cin;
                const int64_t o = ((int64_t)co * Cin + ci) * (KH * KW) + tap;
                out[o] = accumulate ? out[o] + t[j] : t[j];
            }
        } else {
            f32x4* o = (f32x4*)(out + idx);
            *o = accumulate ? *o + t : t;
        }
    }
}

}  // namespace

// Number of pixel splits for a problem (the caller sizes the slab with it).  Workgroups are dealt round-robin to the 8
// XCDs and XCD g runs the tiles of splits g, g+8, ..., so with splits = 8 s every CU receives b = tiles * s / 32 equal
// workgroups and finishes after ceil(b) of them: a launch with b = 4.5 costs as much as b = 5.  More splits also mean a
// larger slab (written once, read once by the reduction).  The count minimises
//     W * occ(b) * ceil(b) / b  +  2 * slab_bytes(s) / 4.5 TB/s  +  1.5 us * ceil(b) ,   W = FLOPs / sustained MFMA rate
// (occ: penalty for resident slots of a CU left empty; the last term is a workgroup's prologue + epilogue), which reproduces the optimum of a measured sweep over the ResNet-18 shapes (tools/wgrad_sweep.py) within ~2 %.
static int resident_per_cu(int BT, bool bf16) {
    // from the kernels' LDS bytes / VGPRs (wave64, 512 VGPRs per SIMD, 160 KB LDS per CU)
    if (bf16) return BT == 64 ? 4 : 3;      // 8-wave workgroups (32 waves per CU); 20-37 / 40-74 KB LDS
    return BT == 64 ? 4 : 2;                // 8-wave workgroups: 32 waves per CU / 64 KB LDS
}

static int choose_splits(int64_t M, int Cin, int Cout, int KH, int KW, bool bf16) {
    const int BT = (Cin <= 64 || Cout <= 64) ? 64 : 128;   // a 128-wide tile would be half empty
    const int64_t tiles = (int64_t)KH * KW * ((Cout + BT - 1) / BT) * ((Cin + BT - 1) / BT);
    static const int target = getenv("SSAD_WGRAD_BLOCKS") ? atoi(getenv("SSAD_WGRAD_BLOCKS")) : 0;
    const int64_t max_splits = (M + 255) / 256;            // at least 256 pixels per workgroup
    if (const char* e = getenv("SSAD_WGRAD_SPLITS")) {     // tuning sweeps (tools/wgrad_sweep.py)
        const int v = atoi(e);
        if (v > 0) return (int)(v > max_splits ? max_splits : v);
    }
    if (target > 0 || max_splits < 16) {                   // explicit target / tiny problems (the head's Linear layers)
        int64_t splits = ((target > 0 ? target : 2048) + tiles - 1) / tiles;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        return (int)(splits > 65535 ? 65535 : splits);
    }
    const double rate = bf16 ? 3.0e14 : (BT == 64 ? 1.05e14 : 1.25e14);            // sustained FLOP/s of the tile kernels
    const double W = 2.0 * (double)M * Cout * KH * KW * Cin / rate;
    const double slab_per_split = 4.0 * (double)Cout * KH * KW * Cin;
    int best_s = 1;
    double best_t = 1e30;
    const int res = resident_per_cu(BT, bf16);
    const double occ_k = BT == 64 ? 0.15 : 0.5;            // cost of CU slots left empty (sweep: l1 +9 % at 2 of 5)
    for (int s = 1; s <= 1024 && (int64_t)s * 8 <= max_splits; ++s) {
        const double bpc = (double)tiles * s / 32.0;        // workgroups per CU
        const double occ = 1.0 + occ_k * (bpc < res ? (res - bpc) / res : 0.0);
        const double t = W * occ * ceil(bpc) / bpc + 2.0 * slab_per_split * 8.0 * s / 4.5e12 + 1.5e-6 * ceil(bpc);
        if (t < best_t * 0.999) { best_t = t; best_s = s; }
    }
    return best_s * 8;
}

extern "C" int ssad_wgrad_splits(int64_t M, int Cin, int Cout, int KH, int KW) {
    return choose_splits(M, Cin, Cout, KH, KW, false);
}

// The same for the bf16-operand kernel (more workgroups resident per CU).
extern "C" int ssad_wgrad_splits_bf16(int64_t M, int Cin, int Cout, int KH, int KW) {
    return choose_splits(M, Cin, Cout, KH, KW, true);
}

static int wgrad_impl(const float* dy, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin,
                               int Cout, int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream, int bf16, int half_in = 0) {
    SSAD_CHECK_ARG(!half_in || bf16 == 2, "half tensors go with fp16 operands");
    SSAD_CHECK_ARG(dy && x && slab, "null pointer");
    SSAD_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "bad shape");
    SSAD_CHECK_ARG(Cin % 4 == 0 && Cout % 4 == 0, "channel counts must be multiples of 4");
    SSAD_CHECK_ARG(splits >= 1 && splits <= 65535, "bad split count");
    WgradParams p;
    p.dy = dy; p.x = x; p.slab = slab;
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    SSAD_CHECK_ARG(p.Ho > 0 && p.Wo > 0, "empty output");
    p.M = N * p.Ho * p.Wo;
    // dy is read over N x Ho x Wo x Cout as derived from x's extents: the caller states what its buffer holds
    SSAD_CHECK_ARG(dy_elems == p.M * Cout, "dy does not hold N x Ho x Wo x Cout elements for this x / filter geometry");
    int64_t chunk = (p.M + splits - 1) / splits;
    p.chunk = (chunk + PK - 1) / PK * PK;
    const int BT = (Cin <= 64 || Cout <= 64) ? 64 : 128;   // a 128-wide tile would be half empty
    p.co_tiles = (Cout + BT - 1) / BT;
    p.ci_tiles = (Cin + BT - 1) / BT;
    p.splits = splits;
    dim3 grid((unsigned)(KH * KW * p.co_tiles * p.ci_tiles * ((splits + 7) / 8) * 8));
    hipStream_t st = (hipStream_t)stream;
    if (bf16 == 6) {
        if (BT == 64) hipLaunchKernelGGL((wgrad_bf16_kernel<64, 3>), grid, dim3(256), 2 * 64 * (3 * PK + 8) * 2, st, p);
        else hipLaunchKernelGGL((wgrad_bf16_kernel<128, 3>), grid, dim3(256), 2 * 128 * (3 * PK + 8) * 2, st, p);
    } else if (bf16 == 3) {
        if (BT == 64) {
            hipLaunchKernelGGL((wgrad_bf16_kernel<64, 2, true>), grid, dim3(512), 2 * 2 * 64 * (2 * PK + 8) * 2, st, p);
        } else {
            static bool x3_attr = false;
            if (!x3_attr) {
                (void)hipFuncSetAttribute((const void*)wgrad_bf16_kernel<128, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          2 * 2 * 128 * (2 * PK + 8) * 2);
                x3_attr = true;
            }
            hipLaunchKernelGGL((wgrad_bf16_kernel<128, 2, true>), grid, dim3(512), 2 * 2 * 128 * (2 * PK + 8) * 2, st, p);
        }
    } else if (bf16 == 2 && half_in) {
        if (BT == 64) hipLaunchKernelGGL((wgrad_bf16_kernel<64, 1, true, true, hf>), grid, dim3(512), 2 * 2 * 64 * (PK + 8) * 2, st, p);
        else hipLaunchKernelGGL((wgrad_bf16_kernel<128, 1, true, true, hf>), grid, dim3(512), 2 * 2 * 128 * (PK + 8) * 2, st, p);
    } else if (bf16 == 2) {
        if (BT == 64) hipLaunchKernelGGL((wgrad_bf16_kernel<64, 1, true, true>), grid, dim3(512), 2 * 2 * 64 * (PK + 8) * 2, st, p);
        else hipLaunchKernelGGL((wgrad_bf16_kernel<128, 1, true, true>), grid, dim3(512), 2 * 2 * 128 * (PK + 8) * 2, st, p);
    } else if (bf16) {
        if (BT == 64) hipLaunchKernelGGL((wgrad_bf16_kernel<64, 1, true>), grid, dim3(512), 2 * 2 * 64 * (PK + 8) * 2, st, p);
        else hipLaunchKernelGGL((wgrad_bf16_kernel<128, 1, true>), grid, dim3(512), 2 * 2 * 128 * (PK + 8) * 2, st, p);
    } else if (BT == 64) {
        hipLaunchKernelGGL(wgrad_f32_kernel<64>, grid, dim3(512), 2 * 2 * PK * 64 * 4, st, p);
    } else {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)wgrad_f32_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      2 * 2 * PK * 128 * 4);
            attr_set = true;
        }
        hipLaunchKernelGGL(wgrad_f32_kernel<128>, grid, dim3(512), 2 * 2 * PK * 128 * 4, st, p);
    }
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_conv_wgrad(const float* dy, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin,
                               int Cout, int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream) {
    return wgrad_impl(dy, x, slab, splits, N, H, W, Cin, Cout, KH, KW, stride, pad, dy_elems, stream, 0);
}

// bf16-operand form (fp32 tensors, fp32 accumulate and slabs): Trainer(precision=16).
extern "C" int ssad_conv_wgrad_bf16(const float* dy, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin,
                                    int Cout, int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream) {
    return wgrad_impl(dy, x, slab, splits, N, H, W, Cin, Cout, KH, KW, stride, pad, dy_elems, stream, 1);
}

// fp16-operand form (the reference's fp16 autocast, tools.py:263); slab sizing as for the bf16 kernel (ssad_wgrad_splits_bf16).
extern "C" int ssad_conv_wgrad_f16(const float* dy, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin,
                                   int Cout, int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream) {
    return wgrad_impl(dy, x, slab, splits, N, H, W, Cin, Cout, KH, KW, stride, pad, dy_elems, stream, 2);
}

// dy and x stored as halves (precision-16 step with half tensors)
extern "C" int ssad_conv_wgrad_f16_h(const void* dy, const void* x, float* slab, int splits, int64_t N, int H, int W, int Cin,
                                     int Cout, int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream) {
    return wgrad_impl((const float*)dy, (const float*)x, slab, splits, N, H, W, Cin, Cout, KH, KW, stride, pad, dy_elems, stream, 2, 1);
}

// split-bf16 ("bf16x3") form: fp32-class accuracy from the bf16 matrix cores (use ssad_wgrad_splits_bf16 for the slab).
// three-way split ("bf16x6"): fp32-faithful products.
extern "C" int ssad_conv_wgrad_x6(const float* dy, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin,
                                  int Cout, int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream) {
    return wgrad_impl(dy, x, slab, splits, N, H, W, Cin, Cout, KH, KW, stride, pad, dy_elems, stream, 6);
}

extern "C" int ssad_conv_wgrad_x3(const float* dy, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin,
                                  int Cout, int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream) {
    return wgrad_impl(dy, x, slab, splits, N, H, W, Cin, Cout, KH, KW, stride, pad, dy_elems, stream, 3);
}

// ---- the slab reductions of SEVERAL weight gradients in one launch (round 6) ----
// Nothing on the backward pass's critical path reads a weight gradient: only the optimizer (and a gradient bucket's all-reduce) does.  The
// training step therefore collects the reductions its weight-gradient kernels leave behind and runs them together where the
// gradients are first needed -- one launch instead of one ~5-12 us launch per layer (19 per step).  Each output is summed exactly as
// wgrad_reduce4_kernel sums it (lane g adds splits g, g + 8, ... in order, then the eight lane sums in order): bit-identical results.
namespace {
struct ReduceTable {
    int n;
    struct E { const float* slab; float* out; int splits, Cout, Kpad, Kreal; int64_t first; } e[24];
};

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(ReduceTable t) {
    __shared__ f32x4 part[8][33];
    int k = 0;
    while (k + 1 < t.n && (int64_t)blockIdx.x >= t.e[k + 1].first) ++k;
    const ReduceTable::E d = t.e[k];
    const int lx = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t idx = (((int64_t)blockIdx.x - d.first) * 32 + lx) * 4;
    const bool ok = idx < (int64_t)d.Cout * d.Kreal;
    const int co = ok ? (int)(idx / d.Kreal) : 0, kk = ok ? (int)(idx - (int64_t)co * d.Kreal) : 0;
    const int64_t stride = (int64_t)d.Cout * d.Kpad;
    const float* s = d.slab + (int64_t)co * d.Kpad + kk;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        int i = g;
        for (; i + 24 < d.splits; i += 32) {
            const f32x4 a = *(const f32x4*)(s + (int64_t)i * stride), b = *(const f32x4*)(s + (int64_t)(i + 8) * stride);
            const f32x4 c = *(const f32x4*)(s + (int64_t)(i + 16) * stride), e = *(const f32x4*)(s + (int64_t)(i + 24) * stride);
            v += a; v += b; v += c; v += e;
        }
        for (; i < d.splits; i += 8) v += *(const f32x4*)(s + (int64_t)i * stride);
    }
    part[g][lx] = v;
    __syncthreads();
    if (g == 0 && ok) {
        f32x4 r = part[0][lx];
#pragma unroll
        for (int q = 1; q < 8; ++q) r += part[q][lx];
        *(f32x4*)(d.out + idx) = r;
    }
}
}  // namespace

// desc[6 k ..]: slab pointer, output pointer (as integers), splits, Cout, Kpad, Kreal = KH * KW * Cin of reduction k: out[co][kk] =
// sum over the splits of slab[s][co][kk] (kk < Kreal; slab rows Kpad floats long).  Kreal % 4 == 0, Kpad % 4 == 0, 16-byte aligned
// pointers (what ssad_wgrad_reduce's vector form asks for); any n (24 reductions per launch).
extern "C" int ssad_wgrad_reduce_batch(const int64_t* desc, int n, void* stream) {
    SSAD_CHECK_ARG(desc && n > 0, "bad argument");
    for (int base = 0; base < n; base += 24) {
        ReduceTable t;
        t.n = n - base < 24 ? n - base : 24;
        int64_t acc = 0;
        for (int k = 0; k < t.n; ++k) {
            const int64_t* d = desc + 6 * (base + k);
            SSAD_CHECK_ARG(d[0] && d[1] && d[2] >= 1 && d[3] > 0 && d[5] > 0 && d[4] >= d[5], "bad reduction descriptor");
            SSAD_CHECK_ARG(d[5] % 4 == 0 && d[4] % 4 == 0 && (d[0] & 15) == 0 && (d[1] & 15) == 0, "vector form: multiples of 4, 16-byte aligned");
            t.e[k].slab = (const float*)(uintptr_t)d[0]; t.e[k].out = (float*)(uintptr_t)d[1];
            t.e[k].splits = (int)d[2]; t.e[k].Cout = (int)d[3]; t.e[k].Kpad = (int)d[4]; t.e[k].Kreal = (int)d[5];
            t.e[k].first = acc;
            acc += cdiv64(d[3] * d[5], 128);
        }
        SSAD_CHECK_ARG(acc < (int64_t)2147483647, "too many blocks");
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)acc), dim3(256), 0, (hipStream_t)stream, t);
        SSAD_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int ssad_wgrad_reduce(const float* slab, float* dw, int splits, int Cout, int Kpad, int KH, int KW, int Cin,
                                 int to_oihw, int accumulate, void* stream) {
    SSAD_CHECK_ARG(slab && dw && splits >= 1 && Cout > 0 && KH > 0 && KW > 0 && Cin > 0, "bad argument");
    SSAD_CHECK_ARG(Kpad >= KH * KW * Cin, "slab rows shorter than the filter");
    const int64_t total = (int64_t)Cout * KH * KW * Cin;
    if ((KH * KW * Cin) % 4 == 0 && Kpad % 4 == 0 && ((uintptr_t)slab & 15) == 0 && ((uintptr_t)dw & 15) == 0)
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)cdiv64(total, 128)), dim3(256), 0, (hipStream_t)stream, slab, dw,
                           splits, Cout, Kpad, KH, KW, Cin, to_oihw, accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(total, 32)), dim3(256), 0, (hipStream_t)stream, slab, dw,
                           splits, Cout, Kpad, KH, KW, Cin, to_oihw, accumulate);
    SSAD_CHECK_LAUNCH();
    return 0;
}
