// Implicit-GEMM convolution / linear layer on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   out[m][co] = act( (sum_k A[m][k] * Wt[co][k]) * scale[co] + shift[co] + residual[m][co] )
//   m = (n, oy, ox) flattened NHWC output pixel, k = (ky, kx, ci) with ci fastest (OHWI weights).
//
// Replaces the nn.Conv2d/BatchNorm2d/ReLU/residual chains of the torchvision BasicBlocks and the
// nn.Linear/BatchNorm1d/ReLU chains of the projection head that the reference runs through cuDNN /
// MKLDNN (src/self_supervised/models.py:224, :247-249).
//
// Design (MI355X): 256-thread workgroups = 4 waves, one per SIMD; each wave owns TM x TN accumulator
// tiles of 32x32 (f32x16 each).  A (gathered activations) and B (weights) K-slices of 32 floats are
// staged global -> registers -> LDS with 144-byte rows (128 B + 16 B pad: conflict-free ds_read_b128
// for the "row = lane&31" fragment pattern), double buffered, one barrier per K-step.  Each lane reads
// 4 consecutive k per ds_read_b128; lane half h takes k = 8*kk + 4*h + e for the e-th MFMA of a chunk,
// the same permutation for A and B, so the contraction is unchanged.  Zero padding is produced by
// predicated loads (no padded copy of the activations exists anywhere).
#include "common.h"

namespace {

constexpr int BK = 32;    // floats per K-step
constexpr int LDK = 36;   // LDS row stride in floats (144 B)

struct ConvParams {
    const float* in;
    const float* wt;
    float* out;
    const float* scale;
    const float* shift;
    const float* residual;
    int64_t M;          // N*Ho*Wo
    int H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, relu;
    int K;              // KH*KW*Cin
    int ts;             // 1: convolution.  >1: transposed gather (dgrad of a stride-ts conv): the tap reads
                        // in[(oy - pad + ky) / ts] only where the numerator is a non-negative multiple of ts
    int posmajor;       // 1: a workgroup's rows are BM different samples at ONE output position, so the set of
                        // in-bounds filter taps is workgroup-uniform and out-of-bounds taps (zero padding) are
                        // skipped as whole K-steps.  Exact: the skipped products are all x*0.
    int64_t N;          // samples (posmajor row bound)
    int hwnc;           // activations (in, out, residual) laid out [H][W][N][C] instead of [N][H][W][C]: with
                        // posmajor rows a workgroup then reads/writes 128 CONSECUTIVE rows of C floats per tap
};

template <int BM, int BN, int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32_kernel(ConvParams p) {
    constexpr int WN = BN / (32 * TN);
    constexpr int AR = BM / 32;     // 16-byte chunks of A staged per thread per K-step
    constexpr int BR = BN / 32;
    constexpr int STAGE = (BM + BN) * LDK;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * BN;
    const int sc = tid & 7, sr = tid >> 3;
    const int HoWo = p.Ho * p.Wo;
    const int ntaps = p.KH * p.KW;

    // ---- row geometry.  normal: row = flattened (n, oy, ox).  posmajor: row = sample, position from blockIdx ----
    int64_t m0;                 // normal: first flattened row; posmajor: first sample
    int pos = 0;
    unsigned long long tapmask = ntaps >= 64 ? ~0ull : ((1ull << ntaps) - 1);
    if (p.posmajor) {
        // Engine-aware mapping.  Workgroups are dealt round-robin to the 8 XCDs and, inside an XCD, to its 4 shader
        // engines: stream = block % 32 is served by one (XCD, SE) pair for the whole launch (measured: with
        // pos = block % HoWo the heavy interior positions pin to the same engines and tap skipping buys nothing).
        // Positions differ in work (corner 4 taps .. interior 9), so every stream sweeps ALL positions of its own
        // sample groups (nb = 32*q + stream): equal work per engine, and a sample group's maps stay in one XCD's L2.
        const int stream = blockIdx.x & 31;
        const int64_t j = blockIdx.x >> 5;
        const int64_t q = j / HoWo;
        pos = (int)(j - q * HoWo);
        const int64_t nb = q * 32 + stream;
        m0 = nb * BM;
        if (m0 >= p.N) return;
        const int oy = pos / p.Wo, ox = pos - oy * p.Wo;
        tapmask = 0;
        for (int t = 0; t < ntaps; ++t) {
            const int ky = t / p.KW, kx = t - ky * p.KW;
            const int y = oy * p.stride - p.pad + ky, x = ox * p.stride - p.pad + kx;
            if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) tapmask |= 1ull << t;
        }
    } else {
        m0 = (int64_t)blockIdx.x * BM;
    }

    // ---- per-thread staging rows (fixed across the K loop) ----
    int64_t a_base[AR];
    int a_iy[AR], a_ix[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int64_t m = m0 + sr + 32 * i;
        if (p.posmajor) {
            const int oy = pos / p.Wo, ox = pos - oy * p.Wo;
            const bool ok = m < p.N;
            a_base[i] = (ok ? m : 0) * (p.hwnc ? (int64_t)p.Cin : (int64_t)p.H * p.W * p.Cin) + sc * 4;
            a_iy[i] = ok ? oy * p.stride - p.pad : -(1 << 20);
            a_ix[i] = ok ? ox * p.stride - p.pad : -(1 << 20);
        } else if (m < p.M) {
            int64_t n = m / HoWo;
            int rem = (int)(m - n * HoWo);
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_base[i] = n * (int64_t)p.H * p.W * p.Cin + sc * 4;
            a_iy[i] = oy * p.stride - p.pad;
            a_ix[i] = ox * p.stride - p.pad;
        } else {
            a_base[i] = 0;
            a_iy[i] = -(1 << 20);
            a_ix[i] = -(1 << 20);
        }
    }
    int64_t b_off[BR];
    bool b_ok[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        int co = n0 + sr + 32 * i;
        b_ok[i] = co < p.Cout;
        b_off[i] = (int64_t)co * p.K + sc * 4;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int64_t in_sp = p.hwnc ? p.N * p.Cin : (int64_t)p.Cin;      // floats between neighbouring pixels
    const int cpt = p.Cin / BK;               // K-steps per filter tap
    const int nk = __builtin_popcountll(tapmask) * cpt;
    f32x4 ra[AR], rb[BR];

    // state of the next K-step to load: current tap (lowest set bit of ld_mask) and channel chunk
    unsigned long long ld_mask = tapmask;
    int ld_tap = tapmask ? __builtin_ctzll(tapmask) : 0;
    int ld_ky = ld_tap / p.KW, ld_kx = ld_tap - ld_ky * p.KW, ld_cc = 0;
    auto load_step = [&]() {
        const int ld_ks = ld_tap * cpt + ld_cc;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            int y = a_iy[i] + ld_ky, x = a_ix[i] + ld_kx;
            bool ok = true;
            if (p.ts > 1) {
                ok = y >= 0 && x >= 0 && (y % p.ts) == 0 && (x % p.ts) == 0;
                y /= p.ts;
                x /= p.ts;
            }
            ok = ok && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4*)(p.in + a_base[i] + ((int64_t)y * p.W + x) * in_sp + ld_cc * BK);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b_ok[i]) v = *(const f32x4*)(p.wt + b_off[i] + (int64_t)ld_ks * BK);
            rb[i] = v;
        }
        if (++ld_cc == cpt) {
            ld_cc = 0;
            ld_mask &= ld_mask - 1;
            ld_tap = ld_mask ? __builtin_ctzll(ld_mask) : 0;
            ld_ky = ld_tap / p.KW;
            ld_kx = ld_tap - ld_ky * p.KW;
        }
    };
    auto store_step = [&](float* buf) {
        float* As = buf;
        float* Bs = buf + BM * LDK;
#pragma unroll
        for (int i = 0; i < AR; ++i) *(f32x4*)(As + (sr + 32 * i) * LDK + sc * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i) *(f32x4*)(Bs + (sr + 32 * i) * LDK + sc * 4) = rb[i];
    };

    if (nk > 0) {
        load_step();
        store_step(lds);
    }
    __syncthreads();

    for (int ks = 0; ks < nk; ++ks) {
        float* cur = lds + (ks & 1) * STAGE;
        const bool more = ks + 1 < nk;
        if (more) load_step();
        const float* As = cur + (wm * 32 * TM + r) * LDK + h * 4;
        const float* Bs = cur + BM * LDK + (wn * 32 * TN + r) * LDK + h * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(As + i * 32 * LDK + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const f32x4*)(Bs + j * 32 * LDK + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i][e], b[j][e], acc[i][j]);
        }
        if (more) store_step(lds + ((ks + 1) & 1) * STAGE);
        __syncthreads();
    }

    // ---- epilogue: affine (folded BN / bias), residual, ReLU; 128-B row segments per half-wave ----
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + r;
        if (col >= p.Cout) continue;
        const float s = p.scale ? p.scale[col] : 1.f;
        const float t = p.shift ? p.shift[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int64_t row = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                bool row_ok = row < p.M;
                if (p.posmajor) {
                    row_ok = row < p.N;
                    row = p.hwnc ? (int64_t)pos * p.N + row : row * HoWo + pos;
                }
                if (row_ok) {
                    int64_t o = row * p.Cout + col;
                    float v = acc[i][j][e] * s + t;
                    if (p.residual) v += p.residual[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.out[o] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int TM, int TN>
int launch(const ConvParams& p, hipStream_t st) {
    constexpr int lds_bytes = 2 * (BM + BN) * LDK * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_f32_kernel<BM, BN, TM, TN>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        attr_set = true;
    }
    const int64_t gx = p.posmajor ? cdiv64(cdiv64(p.N, BM), 32) * 32 * p.Ho * p.Wo : cdiv64(p.M, BM);
    dim3 grid((unsigned)gx, (unsigned)((p.Cout + BN - 1) / BN));
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, TM, TN>), grid, dim3(256), lds_bytes, st, p);
    return 0;
}

}  // namespace

static int conv_fwd_impl(const float* in, const float* w_ohwi, float* out, const float* scale,
                                   const float* shift, const float* residual, int relu, int64_t N, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int stride, int pad, int hwnc, void* stream) {
    SSAD_CHECK_ARG(in && w_ohwi && out, "null pointer");
    SSAD_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "empty shape");
    SSAD_CHECK_ARG(Cin % BK == 0, "Cin must be a multiple of 32");
    SSAD_CHECK_ARG(KH > 0 && KW > 0 && stride > 0 && pad >= 0, "bad filter geometry");
    ConvParams p;
    p.in = in; p.wt = w_ohwi; p.out = out; p.scale = scale; p.shift = shift; p.residual = residual;
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.relu = relu;
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    p.ts = 1;
    SSAD_CHECK_ARG(p.Ho > 0 && p.Wo > 0, "empty output");
    SSAD_CHECK_ARG(KH * KW <= 64, "at most 64 filter taps");
    p.M = N * p.Ho * p.Wo;
    p.N = N;
    p.K = KH * KW * Cin;
    // position-major rows pay off when padding is a visible share of the taps (small maps, many samples)
    p.posmajor = (hwnc || (pad > 0 && N >= 128 && p.Ho * p.Wo <= 4)) ? 1 : 0;
    p.hwnc = hwnc;
    SSAD_CHECK_ARG(cdiv64(p.M, 128) + 32 * p.Ho * p.Wo < (int64_t)2147483647, "M too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    if (Cout <= 64) launch<128, 64, 1, 2>(p, st);
    else launch<128, 128, 2, 2>(p, st);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_conv_igemm_fwd(const float* in, const float* w_ohwi, float* out, const float* scale,
                                   const float* shift, const float* residual, int relu, int64_t N, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream) {
    return conv_fwd_impl(in, w_ohwi, out, scale, shift, residual, relu, N, H, W, Cin, Cout, KH, KW, stride, pad, 0, stream);
}

// Same contraction with every activation tensor (in, out, residual) stored position-major, [H][W][N][C].
// This is the layout of the patch-scoring trunk: N = thousands of 64x64 patches whose maps are 16x16 .. 2x2,
// so a workgroup's 128 rows (128 patches at one output position) are contiguous in HBM for every tap and taps
// that fall into the zero padding are skipped as whole K-steps.
extern "C" int ssad_conv_igemm_fwd_hwnc(const float* in, const float* w_ohwi, float* out, const float* scale,
                                        const float* shift, const float* residual, int relu, int64_t N, int H, int W,
                                        int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream) {
    return conv_fwd_impl(in, w_ohwi, out, scale, shift, residual, relu, N, H, W, Cin, Cout, KH, KW, stride, pad, 1, stream);
}

// dgrad: dx[n][iy][ix][ci] = sum_{ky,kx,co} dy[n][(iy+pad-ky)/s][(ix+pad-kx)/s][co] * w[co][ky][kx][ci] (+ residual).
// w_flipT is ssad_flip_transpose_weight(w): [Cin][KH][KW][Cout] with both taps reversed, so the sum becomes the
// same gather-GEMM with k = (ky', kx', co), numerator row = iy - (KH-1-pad) + ky'.
extern "C" int ssad_conv_igemm_dgrad(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N,
                                     int Hy, int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride,
                                     int pad, void* stream) {
    SSAD_CHECK_ARG(dy && w_flipT && dx, "null pointer");
    SSAD_CHECK_ARG(N > 0 && Hy > 0 && Wy > 0 && Hx > 0 && Wx > 0 && Cin > 0 && Cout > 0, "empty shape");
    SSAD_CHECK_ARG(Cout % BK == 0, "Cout (the contraction) must be a multiple of 32");
    SSAD_CHECK_ARG(KH > 0 && KW > 0 && stride > 0 && pad >= 0 && pad < KH && pad < KW, "bad filter geometry");
    SSAD_CHECK_ARG((Hx + 2 * pad - KH) / stride + 1 == Hy && (Wx + 2 * pad - KW) / stride + 1 == Wy, "dy/dx sizes disagree");
    ConvParams p;
    p.in = dy; p.wt = w_flipT; p.out = dx; p.scale = nullptr; p.shift = nullptr; p.residual = residual;
    p.H = Hy; p.W = Wy; p.Cin = Cout; p.Cout = Cin; p.KH = KH; p.KW = KW; p.relu = 0;
    p.stride = 1; p.pad = KH - 1 - pad; p.ts = stride; p.posmajor = 0; p.N = N; p.hwnc = 0;
    SSAD_CHECK_ARG(KH == KW && KH * KW <= 64, "square filters with at most 64 taps only");
    p.Ho = Hx; p.Wo = Wx;
    p.M = N * Hx * Wx;
    p.K = KH * KW * Cout;
    SSAD_CHECK_ARG(cdiv64(p.M, 128) < (int64_t)2147483647, "M too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    if (Cin <= 64) launch<128, 64, 1, 2>(p, st);
    else launch<128, 128, 2, 2>(p, st);
    SSAD_CHECK_LAUNCH();
    return 0;
}
