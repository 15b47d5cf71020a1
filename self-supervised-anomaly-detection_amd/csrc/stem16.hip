// Stem conv 7x7 / stride 2 / pad 3 (3 -> 64) of the precision-16 training step: fp16 (or bf16) OPERANDS, fp32 accumulation, raw output z
// (NHWC fp32) + the train-mode BatchNorm statistics of bn1 from the accumulators -- the 16-bit counterpart of ssad_stem_fwd_stats
// (stem.hip).  Under the reference's pl.Trainer(precision=16) (tools.py:263) autocast runs resnet.conv1 in fp16 (models.py:224).
//
// The fp32 kernel spends 0.85 ms of a batch-256 step on 84 fp32 MFMA K-steps per tile (0.6 of the matrix peak); with 16-bit operands
// the matrix work is 1/12 of that and the kernel is bound by writing z (1.07 GB).  v_mfma_f32_32x32x16_f16 wants, per lane, 8
// consecutive K values: the input tile lives in LDS as halves [row][col][4] (channel 3 = 0), and for a filter row ky the K index is
// k = 4 kx + c with kx padded 7 -> 8 (zero weights): K = 7 x 32 = 224 = 14 MFMA K-steps.  Output pixel ox of a tile row reads
// columns 2 ox .. 2 ox + 7: lane (r = ox, h) takes k = 16 kk + 8 h .. + 7, i.e. the 16 bytes at column 2 ox + 4 kk + 2 h -- always
// 16-byte aligned.  Weights: wk16[ky][kk][co][16] halves (ssad_pack_stem_weight16), all 28 KB resident in LDS.
// Accumulator layout = the fp32 kernel's (rows = pixels, columns = channels), so statistics and stores are the same code.
#include "common.h"

namespace {

constexpr int S16_MAX_GRID = 4096;           // persistent workgroups = rows of the statistics partials (ssad_stem_stats_rows)
constexpr int TOH = 8, TOW = 32;             // output tile per workgroup iteration
constexpr int TIH = TOH * 2 + 5;             // 21 input rows
constexpr int TIW = 72;                      // >= 2 TOW + 5 + 3 (kx pad + the h half) and even
constexpr int IN_H = TIH * TIW * 4;          // halves
constexpr int W_H = 14 * 64 * 16;            // halves

struct Stem16Params {
    const float* img;     // [B][3][H][W]
    const void* wk16;     // [7][2][64][16] halves
    void* out;            // [B][Ho][Wo][64] floats, or halves (TO = hf: the precision-16 step keeps its activations as halves)
    double* stats;        // [gridDim.x][2][64]
    int B, H, W, Ho, Wo, tiles_y, tiles_x;
    int64_t total_tiles;
};

template <bool F16> struct Op16;
template <> struct Op16<false> {
    using t = __bf16; using v4 = bf16x4; using v8 = bf16x8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Op16<true> {
    using t = _Float16; using v4 = f16x4; using v8 = f16x8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <bool F16, typename TO = float>
__global__ __launch_bounds__(256, 2) void stem_conv7x7_16_kernel(Stem16Params p) {
    TO* const outp = (TO*)p.out;
    using op_t = typename Op16<F16>::t;
    using op4 = typename Op16<F16>::v4;
    using op8 = typename Op16<F16>::v8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    op_t* wl = (op_t*)lds;               // [14][64][16]
    op_t* tin = wl + W_H;                // [TIH][TIW][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    for (int i = tid; i < W_H / 8; i += 256) ((f32x4*)wl)[i] = ((const f32x4*)p.wk16)[i];

    double st0[2] = {0.0, 0.0}, st1[2] = {0.0, 0.0};      // train-mode BatchNorm statistics of this lane's channels
    const int tiles_per_sample = p.tiles_y * p.tiles_x;
    const int64_t plane = (int64_t)p.H * p.W;
    // the (zero-padded) input tile of the workgroup's NEXT tile is fetched into registers while the current tile is computed
    constexpr int NIN = (TIH * TIW + 255) / 256;             // 6 pixels per thread
    float pv[NIN][3];
    auto fetch_tile = [&](int64_t t) {
        const int64_t n = t / tiles_per_sample;
        const int tt = (int)(t - n * tiles_per_sample);
        const int ty0 = (tt / p.tiles_x) * TOH, tx0 = (tt % p.tiles_x) * TOW;
        const float* src = p.img + n * 3 * plane;
#pragma unroll
        for (int q = 0; q < NIN; ++q) {
            const int i = tid + 256 * q;
            const int iy = i / TIW, ix = i - iy * TIW;
            const int vy = 2 * ty0 - 3 + iy, vx = 2 * tx0 - 3 + ix;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (i < TIH * TIW && (unsigned)vy < (unsigned)p.H && (unsigned)vx < (unsigned)p.W) {
                const float* s = src + (int64_t)vy * p.W + vx;
                v0 = s[0]; v1 = s[plane]; v2 = s[2 * plane];
            }
            pv[q][0] = v0; pv[q][1] = v1; pv[q][2] = v2;
        }
    };
    if ((int64_t)blockIdx.x < p.total_tiles) fetch_tile(blockIdx.x);
    for (int64_t t = blockIdx.x; t < p.total_tiles; t += gridDim.x) {
        const int64_t n = t / tiles_per_sample;
        const int tt = (int)(t - n * tiles_per_sample);
        const int ty0 = (tt / p.tiles_x) * TOH, tx0 = (tt % p.tiles_x) * TOW;

        __syncthreads();   // previous tile's readers are done with tin (and wl is visible on first pass)
#pragma unroll
        for (int q = 0; q < NIN; ++q) {
            const int i = tid + 256 * q;
            if (i < TIH * TIW) {
                const op4 v = {(op_t)pv[q][0], (op_t)pv[q][1], (op_t)pv[q][2], (op_t)0.f};
                *(op4*)(tin + i * 4) = v;
            }
        }
        __syncthreads();
        if (t + gridDim.x < p.total_tiles) fetch_tile(t + gridDim.x);

        // wave w computes output rows ty0 + 2w, ty0 + 2w + 1 (32 pixels each) x 64 channels
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const op_t* a0 = tin + ((2 * (2 * wave + 0)) * TIW + 2 * r + 2 * h) * 4;
        const op_t* a1 = tin + ((2 * (2 * wave + 1)) * TIW + 2 * r + 2 * h) * 4;
        const op_t* bw = wl + r * 16 + 8 * h;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const op8 x0v = *(const op8*)(a0 + (ky * TIW + 4 * kk) * 4);
                const op8 x1v = *(const op8*)(a1 + (ky * TIW + 4 * kk) * 4);
                const op8 w0 = *(const op8*)(bw + (ky * 2 + kk) * 64 * 16);
                const op8 w1 = *(const op8*)(bw + (ky * 2 + kk) * 64 * 16 + 32 * 16);
                acc[0][0] = Op16<F16>::mfma(x0v, w0, acc[0][0]);
                acc[0][1] = Op16<F16>::mfma(x0v, w1, acc[0][1]);
                acc[1][0] = Op16<F16>::mfma(x1v, w0, acc[1][0]);
                acc[1][1] = Op16<F16>::mfma(x1v, w1, acc[1][1]);
            }

        // ---- epilogue (stem.hip's): statistics in float per tile row, one double add per row; raw z stored ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = ty0 + 2 * wave + i;
            if (oy >= p.Ho) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float fs = 0.f, fq = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ox = tx0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (ox < p.Wo) {
                        const float v = stored<TO>(acc[i][j][e]);          // statistics of what is stored
                        fs += v; fq += v * v;
                        const int64_t pix = (n * p.Ho + oy) * p.Wo + ox;
                        outp[pix * 64 + j * 32 + r] = (TO)v;
                    }
                }
                st0[j] += (double)fs; st1[j] += (double)fq;
            }
        }
    }
    // lane halves -> one value per channel per wave, the four waves in a fixed order through LDS (the tiles are dead)
    __syncthreads();
    double* S = (double*)lds;                    // [4 waves][2][64]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        st0[j] += __shfl_xor(st0[j], 32);
        st1[j] += __shfl_xor(st1[j], 32);
        if (h == 0) {
            S[(wave * 2 + 0) * 64 + j * 32 + r] = st0[j];
            S[(wave * 2 + 1) * 64 + j * 32 + r] = st1[j];
        }
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, cc = tid & 63;
        p.stats[((int64_t)blockIdx.x * 2 + which) * 64 + cc] =
            ((S[(0 * 2 + which) * 64 + cc] + S[(1 * 2 + which) * 64 + cc]) + S[(2 * 2 + which) * 64 + cc]) + S[(3 * 2 + which) * 64 + cc];
    }
}

// OIHW [64][3][7][7] fp32 -> [ky][kk][co][16] halves, k = 16 kk + j <-> (kx = (16 kk + j) / 4, c = j % 4); kx = 7 and c = 3 are zeros
template <bool F16>
__global__ void pack_stem_weight16_kernel(const float* __restrict__ w, typename Op16<F16>::t* __restrict__ wk, int ohwi = 0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W_H) return;
    const int j = i & 15, co = (i >> 4) & 63, kk = (i >> 10) & 1, ky = i >> 11;
    const int k = 16 * kk + j, kx = k >> 2, c = k & 3;
    const bool real = kx < 7 && c < 3;
    const int kxc = real ? kx : 0, cc = real ? c : 0;
    const float t = w[ohwi ? ((co * 7 + ky) * 7 + kxc) * 3 + cc : ((co * 3 + cc) * 7 + ky) * 7 + kxc];      // ohwi: [64][7][7][3]
    wk[i] = (typename Op16<F16>::t)(real ? t : 0.f);
}

}  // namespace

// wk16: 14 * 64 * 16 halves (28 672 bytes); f16 != 0: fp16, else bf16
static int pack16(const float* w, void* wk16, int f16, int ohwi, void* stream) {
    SSAD_CHECK_ARG(w && wk16, "null pointer");
    if (f16) hipLaunchKernelGGL(pack_stem_weight16_kernel<true>, dim3((W_H + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)wk16, ohwi);
    else hipLaunchKernelGGL(pack_stem_weight16_kernel<false>, dim3((W_H + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)wk16, ohwi);
    SSAD_CHECK_LAUNCH();
    return 0;
}
extern "C" int ssad_pack_stem_weight16(const float* w_oihw, void* wk16, int f16, void* stream) { return pack16(w_oihw, wk16, f16, 0, stream); }
// ... from an OHWI filter [64][7][7][3] (the parameter arena's layout; see ssad_pack_stem_weight_ohwi)
extern "C" int ssad_pack_stem_weight16_ohwi(const float* w_ohwi, void* wk16, int f16, void* stream) { return pack16(w_ohwi, wk16, f16, 1, stream); }

// The 16-bit-operand form of ssad_stem_fwd_stats (whole images, no patch windows): z [B][Ho][Wo][64] fp32, mean / invstd / running
// statistics of bn1; workspace: ssad_stem_stats_rows() * 128 doubles.
static int stem_fwd_stats16_impl(const float* img, int B, int H, int W, const void* wk16, void* out, int out_half, float eps, float momentum,
                                 float* mean, float* invstd, float* running_mean, float* running_var, double* workspace, int f16,
                                 void* stream) {
    SSAD_CHECK_ARG(img && wk16 && out && mean && invstd && workspace, "null pointer");
    SSAD_CHECK_ARG(B > 0 && H >= 64 && W >= 64, "whole images of at least 64 x 64 (smaller ones are resized first: ssad_stem_fwd_stats)");
    Stem16Params p;
    p.img = img; p.wk16 = wk16; p.out = out; p.stats = workspace; p.B = B; p.H = H; p.W = W;
    p.Ho = (H - 1) / 2 + 1; p.Wo = (W - 1) / 2 + 1;
    p.tiles_y = (p.Ho + TOH - 1) / TOH; p.tiles_x = (p.Wo + TOW - 1) / TOW;
    p.total_tiles = (int64_t)B * p.tiles_y * p.tiles_x;
    constexpr int lds_bytes = (W_H + IN_H) * 2;
    const int64_t grid = p.total_tiles < S16_MAX_GRID ? p.total_tiles : S16_MAX_GRID;
    if (out_half) {
        SSAD_CHECK_ARG(f16, "half output goes with fp16 operands");
        hipLaunchKernelGGL((stem_conv7x7_16_kernel<true, hf>), dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, p);
    } else if (f16) hipLaunchKernelGGL(stem_conv7x7_16_kernel<true>, dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(stem_conv7x7_16_kernel<false>, dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, p);
    SSAD_CHECK_LAUNCH();
    return ssad_bn_finalize_partials(workspace, (int)grid, (int64_t)B * p.Ho * p.Wo, 64, eps, momentum, mean, invstd, running_mean,
                                     running_var, stream);
}

extern "C" int ssad_stem_fwd_stats16(const float* img, int B, int H, int W, const void* wk16, float* out, float eps, float momentum,
                                     float* mean, float* invstd, float* running_mean, float* running_var, double* workspace, int f16,
                                     void* stream) {
    return stem_fwd_stats16_impl(img, B, H, W, wk16, out, 0, eps, momentum, mean, invstd, running_mean, running_var, workspace, f16, stream);
}

// fp16 operands and z stored as halves [B][Ho][Wo][64]; the statistics are those of the stored halves
extern "C" int ssad_stem_fwd_stats16_h(const float* img, int B, int H, int W, const void* wk16, void* out, float eps, float momentum,
                                       float* mean, float* invstd, float* running_mean, float* running_var, double* workspace,
                                       void* stream) {
    return stem_fwd_stats16_impl(img, B, H, W, wk16, out, 1, eps, momentum, mean, invstd, running_mean, running_var, workspace, 1, stream);
}
