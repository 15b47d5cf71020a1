// Winograd F(2x2, 3x3) convolution (stride 1, pad 1) for the patch-scoring trunk, position-major layout [H][W][N][C].
//
// Replaces the same call sites as ssad_conv_igemm_fwd_hwnc for 3x3/1 convs (the Conv2d + eval BatchNorm2d + residual
// + ReLU of the torchvision BasicBlocks, src/self_supervised/models.py:224) with 4 instead of 9 multiplies per output:
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// fp32 throughout; the result differs from the direct sum only by rounding (a few ulp of the accumulated magnitude;
// parity tests hold it to the same 1e-4 bound as the direct kernel).
//
// Mapping: a workgroup owns BM samples x BN output channels at ONE 2x2 output tile.  It runs 16 GEMM segments
// (xi, nu) over Cin on the fp32 matrix cores; the input transform B^T d B is applied while staging (each transformed
// row is +-d00 +-d01 +-d10 +-d11 of four input pixels, read as contiguous rows of the [H][W][N][C] tensor; pixels in
// the zero padding come from a zero page), the output transform is folded into four accumulator sets as each
// segment finishes (coefficients 0/+-1), and the epilogue writes the four output pixels through LDS in 16-byte pieces.
// Every tile position carries identical work, so no load balancing is needed (cf. conv_igemm.hip).
//
// Measured (round 1, 8192 patches): 8-9 % faster than the tap-skipping direct kernel on 8x8 and 4x4 maps, on par at
// 16x16, slower at 2x2 (where tap skipping already leaves 4 of 9 taps).  The 2.25x MAC saving does not convert:
// the accumulator sets (M + 4 Y) leave one wave per SIMD and the staged transform quadruples L2 traffic.  An
// "A-stationary" variant (raw patch pixels in LDS, transform at fragment-read time) was built and measured slower
// (25 % MFMA utilisation: ds_read -> fma -> mfma chains with nothing to overlap), and removed.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int WBK = 32;
constexpr int WLDK = 36;

__device__ __attribute__((aligned(16))) float g_wino_zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};

struct WinoParams {
    const float* in;
    const float* u;        // transformed weights [16][Cout][Cin]
    float* out;
    const float* scale;
    const float* shift;
    const float* residual;
    int64_t N;
    int H, W, Cin, Cout, relu;
};

// rows of B^T: xi -> the two input rows it combines and their signs
__device__ __forceinline__ void bt_row(int xi, int& i0, int& i1, float& s0, float& s1) {
    i0 = xi == 0 ? 0 : 1;
    i1 = xi == 3 ? 3 : 2;
    s0 = xi == 2 ? -1.f : 1.f;
    s1 = (xi == 0 || xi == 3) ? -1.f : 1.f;
}
// A^T = [[1,1,1,0],[0,1,-1,-1]]
__device__ __forceinline__ float at_coef(int a, int xi) {
    return a == 0 ? (xi == 3 ? 0.f : 1.f) : (xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f));
}

template <int TM, int TN>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_hwnc_kernel(WinoParams p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;       // waves 2 x 2, wave tile (32 TM) x (32 TN)
    constexpr int AR = BM / 32, BR = BN / 32;
    constexpr int STAGE = (BM + BN) * WLDK;
    constexpr int LDC = BN + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int sc = tid & 7, sr = tid >> 3;
    const int n0 = blockIdx.y * BN;
    const int TW = p.W >> 1, NT = (p.H >> 1) * TW;
    const float* zero = g_wino_zero;

    const int stream = blockIdx.x & 31;                 // (XCD, SE) stream, see conv_igemm.hip
    const int64_t j = blockIdx.x >> 5;
    const int64_t q = j / NT;
    const int tp = (int)(j - q * NT);
    const int64_t m0 = (q * 32 + stream) * BM;
    if (m0 >= p.N) return;
    const int ty = tp / TW, tx = tp - ty * TW;
    const int64_t in_sp = p.N * p.Cin;                  // floats between neighbouring pixels

    const float* a_ptr[AR];
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int64_t n = m0 + sr + 32 * i;
        a_ok[i] = n < p.N;
        a_ptr[i] = p.in + (a_ok[i] ? n : 0) * p.Cin + sc * 4;
    }
    const float* b_ptr[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int co = n0 + sr + 32 * i;
        b_ptr[i] = co < p.Cout ? p.u + (int64_t)co * p.Cin + sc * 4 : nullptr;
    }

    f32x16 accM[TM][TN], accY[4][TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                accM[i][jn][e] = 0.f;
#pragma unroll
                for (int o = 0; o < 4; ++o) accY[o][i][jn][e] = 0.f;
            }

    const int cpt = p.Cin / WBK;
    const int nk = 16 * cpt;
    f32x4 raw[AR][4], rb[BR];
    float cf[4];                                        // signs of the 4 pixels for the step held in raw[]

    int ld_g = 0, ld_cc = 0;
    auto load_step = [&]() {
        const int xi = ld_g >> 2, nu = ld_g & 3;
        int i0, i1, j0, j1;
        float si0, si1, sj0, sj1;
        bt_row(xi, i0, i1, si0, si1);
        bt_row(nu, j0, j1, sj0, sj1);
        const int y0 = 2 * ty - 1 + i0, y1 = 2 * ty - 1 + i1, x0 = 2 * tx - 1 + j0, x1 = 2 * tx - 1 + j1;
        const bool vy0 = (unsigned)y0 < (unsigned)p.H, vy1 = (unsigned)y1 < (unsigned)p.H;
        const bool vx0 = (unsigned)x0 < (unsigned)p.W, vx1 = (unsigned)x1 < (unsigned)p.W;
        const int64_t cco = (int64_t)ld_cc * WBK;
        const int64_t o00 = ((int64_t)y0 * p.W + x0) * in_sp + cco, o01 = ((int64_t)y0 * p.W + x1) * in_sp + cco;
        const int64_t o10 = ((int64_t)y1 * p.W + x0) * in_sp + cco, o11 = ((int64_t)y1 * p.W + x1) * in_sp + cco;
        cf[0] = si0 * sj0; cf[1] = si0 * sj1; cf[2] = si1 * sj0; cf[3] = si1 * sj1;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            raw[i][0] = *(const f32x4*)((a_ok[i] && vy0 && vx0) ? a_ptr[i] + o00 : zero);
            raw[i][1] = *(const f32x4*)((a_ok[i] && vy0 && vx1) ? a_ptr[i] + o01 : zero);
            raw[i][2] = *(const f32x4*)((a_ok[i] && vy1 && vx0) ? a_ptr[i] + o10 : zero);
            raw[i][3] = *(const f32x4*)((a_ok[i] && vy1 && vx1) ? a_ptr[i] + o11 : zero);
        }
        const int64_t woff = (int64_t)ld_g * p.Cout * p.Cin + cco;
#pragma unroll
        for (int i = 0; i < BR; ++i) rb[i] = *(const f32x4*)(b_ptr[i] ? b_ptr[i] + woff : zero);
        if (++ld_cc == cpt) { ld_cc = 0; ++ld_g; }
    };
    auto store_step = [&](float* buf) {
        float* As = buf;
        float* Bs = buf + BM * WLDK;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = cf[0] * raw[i][0][k] + cf[1] * raw[i][1][k] + cf[2] * raw[i][2][k] + cf[3] * raw[i][3][k];
            *(f32x4*)(As + (sr + 32 * i) * WLDK + sc * 4) = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) *(f32x4*)(Bs + (sr + 32 * i) * WLDK + sc * 4) = rb[i];
    };

    load_step();
    store_step(lds);
    __syncthreads();

    int cg = 0, ccc = 0;                                // segment / chunk of the step being computed
    for (int ks = 0; ks < nk; ++ks) {
        float* cur = lds + (ks & 1) * STAGE;
        const bool more = ks + 1 < nk;
        if (more) load_step();
        const float* As = cur + (wm * 32 * TM + r) * WLDK + h * 4;
        const float* Bs = cur + BM * WLDK + (wn * 32 * TN + r) * WLDK + h * 4;
#pragma unroll
        for (int kk = 0; kk < WBK / 8; ++kk) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(As + i * 32 * WLDK + kk * 8);
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) b[jn] = *(const f32x4*)(Bs + jn * 32 * WLDK + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) accM[i][jn] = mfma32(a[i][e], b[jn][e], accM[i][jn]);
        }
        if (++ccc == cpt) {
            // segment (xi, nu) finished: fold M into the four output accumulators, Y[a][b] += A^T[a][xi] A^T[b][nu] M
            const int xi = cg >> 2, nu = cg & 3;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const float c = at_coef(o >> 1, xi) * at_coef(o & 1, nu);
                if (c != 0.f) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                            for (int e = 0; e < 16; ++e) accY[o][i][jn][e] += c * accM[i][jn][e];
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                    for (int e = 0; e < 16; ++e) accM[i][jn][e] = 0.f;
            ccc = 0;
            ++cg;
        }
        if (more) store_step(lds + ((ks + 1) & 1) * STAGE);
        __syncthreads();
    }

    // ---- epilogue: for each of the 4 output pixels, accumulator tile -> LDS -> 16-byte pieces ----
    float* C = lds;
    constexpr int EM = BM / TM;                // rows per pass (one accumulator row-tile per wave-row)
    constexpr int C4 = BN / 4, RP = 256 / C4, NPASS = EM / RP;
    const int c4 = tid % C4, rr = tid / C4;
    const int col = n0 + c4 * 4;
    const bool col_ok = col < p.Cout;
    f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
    if (col_ok && p.scale) s4 = *(const f32x4*)(p.scale + col);
    if (col_ok && p.shift) t4 = *(const f32x4*)(p.shift + col);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int64_t pos = (int64_t)(2 * ty + (o >> 1)) * p.W + (2 * tx + (o & 1));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            __syncthreads();
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    C[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + (wn * TN + jn) * 32 + r] = accY[o][i][jn][e];
            __syncthreads();
            int64_t off[NPASS];
            f32x4 res[NPASS];
#pragma unroll
            for (int qq = 0; qq < NPASS; ++qq) {
                const int lr = rr + qq * RP;
                const int64_t n = m0 + ((lr >> 5) * TM + i) * 32 + (lr & 31);
                const bool ok = col_ok && n < p.N;
                off[qq] = ok ? (pos * p.N + n) * p.Cout + col : -1;
                res[qq] = *(const f32x4*)((ok && p.residual) ? p.residual + off[qq] : zero);
            }
#pragma unroll
            for (int qq = 0; qq < NPASS; ++qq) {
                f32x4 v = *(const f32x4*)(C + (rr + qq * RP) * LDC + c4 * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float x = v[k] * s4[k] + t4[k] + res[qq][k];
                    v[k] = p.relu ? fmaxf(x, 0.f) : x;
                }
                if (off[qq] >= 0) *(f32x4*)(p.out + off[qq]) = v;
            }
        }
    }
}

// OHWI [Cout][3][3][Cin] -> U[xi*4+nu][Cout][Cin] = (G g G^T)[xi][nu],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Cout * Cin) return;
    const int ci = (int)(idx % Cin), co = (int)(idx / Cin);
    float g[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = w[(((int64_t)co * 3 + a) * 3 + b) * Cin + ci];
    float t[4][3];
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    for (int a = 0; a < 4; ++a) {
        const float uu[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
        for (int b = 0; b < 4; ++b) u[((int64_t)(a * 4 + b) * Cout + co) * Cin + ci] = uu[b];
    }
}

template <int TM, int TN>
void launch_wino(const WinoParams& p, hipStream_t st) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int stage_bytes = 2 * (BM + BN) * WLDK * 4;
    constexpr int epi_bytes = (BM / TM) * (BN + 4) * 4;
    constexpr int lds_bytes = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_wino_hwnc_kernel<TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        attr_set = true;
    }
    const int64_t NT = (int64_t)(p.H / 2) * (p.W / 2);
    const int64_t gx = cdiv64(cdiv64(p.N, BM), 32) * 32 * NT;
    dim3 grid((unsigned)gx, (unsigned)((p.Cout + BN - 1) / BN));
    hipLaunchKernelGGL((conv3x3_wino_hwnc_kernel<TM, TN>), grid, dim3(256), lds_bytes, st, p);
}

}  // namespace

extern "C" int ssad_wino_weight_transform(const float* w_ohwi, float* u, int Cout, int Cin, void* stream) {
    SSAD_CHECK_ARG(w_ohwi && u && Cout > 0 && Cin > 0, "bad argument");
    const int64_t total = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w_ohwi, u, Cout, Cin);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_conv3x3_wino_fwd_hwnc(const float* in, const float* u, float* out, const float* scale, const float* shift,
                                          const float* residual, int relu, int64_t N, int H, int W, int Cin, int Cout,
                                          void* stream) {
    SSAD_CHECK_ARG(in && u && out, "null pointer");
    SSAD_CHECK_ARG(N > 0 && H >= 2 && W >= 2 && (H % 2) == 0 && (W % 2) == 0, "even map sizes only");
    SSAD_CHECK_ARG(Cin % WBK == 0 && Cout % 4 == 0, "Cin % 32 and Cout % 4");
    WinoParams p;
    p.in = in; p.u = u; p.out = out; p.scale = scale; p.shift = shift; p.residual = residual;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
    SSAD_CHECK_ARG(cdiv64(N, 64) * (H / 2) * (W / 2) + 32 * (H / 2) * (W / 2) < (int64_t)2147483647, "too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    if (Cout <= 64) launch_wino<2, 1>(p, st);
    else launch_wino<2, 2>(p, st);
    SSAD_CHECK_LAUNCH();
    return 0;
}
