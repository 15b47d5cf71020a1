// Stem: sliding-window patch extraction + nearest resize + conv 7x7 stride 2 pad 3 (3 -> 64) + affine + ReLU,
// then the 3x3/2 max-pool, for NCHW fp32 images.
//
// Replaces extract_patches (src/self_supervised/functional.py:77-82), the (b*p,c,h,w) reshape copy and
// F.interpolate(x, 64, 'nearest') (src/self_supervised/models.py:212-219) and resnet conv1/bn1/relu/maxpool
// (models.py:224).  The patch window and the resize are index arithmetic in the tile loader: sample
// n = b*P + pr*ncols + pc reads source pixel (ps*pr + floor(vy*ph/Hv), ps*pc + floor(vx*pw/Wv)).
//
// MFMA formulation: out[pixel][co] = sum_{ky,q,c,h} in[2oy+ky-3][2ox+2q+h-3][c] * Wk[((ky*4+q)*3+c)*2+h][co]
// (kx = 2q+h padded 7 -> 8 with a zero weight row) so that the two lane halves of v_mfma_f32_32x32x2_f32
// read LDS addresses a constant 3 floats apart and every K-step offset is an immediate.
#include "common.h"

namespace {

constexpr int TOH = 8, TOW = 32;             // output tile per workgroup iteration
constexpr int TIH = TOH * 2 + 5;             // 21 input rows
constexpr int TIW = 72;                      // >= TOW*2 + 5 + 1 (kx pad) = 70
constexpr int KSTEPS = 7 * 4 * 3;            // 84 MFMA K-steps (K = 168)
constexpr int IN_TILE = TIH * TIW * 3;       // floats
constexpr int W_TILE = KSTEPS * 2 * 64;      // 10752 floats

struct StemParams {
    const float* img;
    const float* wk;
    const float* scale;
    const float* shift;
    float* out;
    int B, H, W, pd, ps, Hv, Wv, Ho, Wo, relu;
    int prow, pcol;       // patches per column / row direction (1,1 in image mode)
    int ph, pw;           // window size in source pixels
    int tiles_y, tiles_x;
    int64_t total_tiles;
    int64_t Nsamp;
    int hwnc;             // write [Ho][Wo][Nsamp][64] instead of [Nsamp][Ho][Wo][64]
};

__global__ __launch_bounds__(256, 2) void stem_conv7x7_kernel(StemParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                 // [168][64]
    float* tin = lds + W_TILE;       // [TIH][TIW][3]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    for (int i = tid; i < W_TILE / 4; i += 256) ((f32x4*)wl)[i] = ((const f32x4*)p.wk)[i];

    float sc[2], sh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        sc[j] = p.scale ? p.scale[j * 32 + r] : 1.f;
        sh[j] = p.shift ? p.shift[j * 32 + r] : 0.f;
    }

    const int tiles_per_sample = p.tiles_y * p.tiles_x;
    for (int64_t t = blockIdx.x; t < p.total_tiles; t += gridDim.x) {
        const int64_t n = t / tiles_per_sample;
        const int tt = (int)(t - n * tiles_per_sample);
        const int ty0 = (tt / p.tiles_x) * TOH, tx0 = (tt % p.tiles_x) * TOW;
        const int P = p.prow * p.pcol;
        const int b = (int)(n / P);
        const int pi = (int)(n - (int64_t)b * P);
        const int y0 = (pi / p.pcol) * p.ps, x0 = (pi % p.pcol) * p.ps;
        const float* src = p.img + (int64_t)b * 3 * p.H * p.W;

        __syncthreads();   // previous tile's readers are done with tin (and wl is visible on first pass)
        // ---- load the (virtual, zero-padded) input tile: rows 2*ty0-3 .., cols 2*tx0-3 .. ----
        for (int i = tid; i < TIH * TIW; i += 256) {
            int iy = i / TIW, ix = i - iy * TIW;
            int vy = 2 * ty0 - 3 + iy, vx = 2 * tx0 - 3 + ix;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if ((unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv) {
                int sy = y0 + (vy * p.ph) / p.Hv, sx = x0 + (vx * p.pw) / p.Wv;
                const float* s = src + (int64_t)sy * p.W + sx;
                int64_t plane = (int64_t)p.H * p.W;
                v0 = s[0]; v1 = s[plane]; v2 = s[2 * plane];
            }
            float* d = tin + i * 3;
            d[0] = v0; d[1] = v1; d[2] = v2;
        }
        __syncthreads();

        // wave w computes output rows ty0 + 2w, ty0 + 2w + 1 (32 pixels each) x 64 channels
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const float* a0 = tin + ((2 * (2 * wave + 0)) * TIW + 2 * r + h) * 3;
        const float* a1 = tin + ((2 * (2 * wave + 1)) * TIW + 2 * r + h) * 3;
        const float* bw = wl + h * 64 + r;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int ko = (ky * TIW + 2 * q) * 3 + c;
                    const int kb = ((ky * 4 + q) * 3 + c) * 128;
                    float x0v = a0[ko], x1v = a1[ko];
                    float w0 = bw[kb], w1 = bw[kb + 32];
                    acc[0][0] = mfma32(x0v, w0, acc[0][0]);
                    acc[0][1] = mfma32(x0v, w1, acc[0][1]);
                    acc[1][0] = mfma32(x1v, w0, acc[1][0]);
                    acc[1][1] = mfma32(x1v, w1, acc[1][1]);
                }

        // ---- epilogue ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = ty0 + 2 * wave + i;
            if (oy >= p.Ho) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int ox = tx0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (ox < p.Wo) {
                        float v = acc[i][j][e] * sc[j] + sh[j];
                        if (p.relu) v = fmaxf(v, 0.f);
                        const int64_t pix = p.hwnc ? ((int64_t)oy * p.Wo + ox) * p.Nsamp + n : (n * p.Ho + oy) * p.Wo + ox;
                        p.out[pix * 64 + j * 32 + r] = v;
                    }
                }
            }
        }
    }
}

__global__ void pack_stem_weight_kernel(const float* __restrict__ w, float* __restrict__ wk) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;     // over 168*64
    if (i >= 168 * 64) return;
    int co = i & 63, kl = i >> 6;
    int hh = kl & 1, rest = kl >> 1;
    int c = rest % 3, q = (rest / 3) & 3, ky = rest / 12;
    int kx = 2 * q + hh;
    wk[i] = kx < 7 ? w[((co * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
}

// NHWC 3x3 stride-2 pad-1 max-pool, 4 channels per thread.
__global__ void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t total4, int H, int W,
                                    int C4, int Ho, int Wo, int64_t N, int hwnc) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    int c4 = (int)(i % C4);
    int64_t pix = i / C4;
    int ox, oy;
    int64_t n;
    if (hwnc) {                       // i enumerates [Ho][Wo][N][C4]
        n = pix % N;
        int64_t t = pix / N;
        ox = (int)(t % Wo);
        oy = (int)(t / Wo);
    } else {
        ox = (int)(pix % Wo);
        int64_t t = pix / Wo;
        oy = (int)(t % Ho);
        n = t / Ho;
    }
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        int y = oy * 2 - 1 + dy;
        if ((unsigned)y >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            int x = ox * 2 - 1 + dx;
            if ((unsigned)x >= (unsigned)W) continue;
            const int64_t ip = hwnc ? ((int64_t)y * W + x) * N + n : (n * H + y) * W + x;
            f32x4 v = ((const f32x4*)in)[ip * C4 + c4];
            m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
    }
    ((f32x4*)out)[i] = m;
}

}  // namespace

extern "C" int ssad_pack_stem_weight(const float* w_oihw, float* wk, void* stream) {
    SSAD_CHECK_ARG(w_oihw && wk, "null pointer");
    hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((168 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, wk);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_stem_fwd(const float* img, int B, int H, int W, int patch_dim, int patch_stride, int Hv, int Wv,
                             const float* wk, const float* scale, const float* shift, int relu, int hwnc, float* out,
                             void* stream) {
    SSAD_CHECK_ARG(img && wk && out, "null pointer");
    SSAD_CHECK_ARG(B > 0 && H > 0 && W > 0 && Hv > 0 && Wv > 0, "empty shape");
    StemParams p;
    p.img = img; p.wk = wk; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.pd = patch_dim; p.ps = patch_stride; p.Hv = Hv; p.Wv = Wv; p.relu = relu;
    if (patch_dim > 0) {
        SSAD_CHECK_ARG(patch_stride > 0 && patch_dim <= H && patch_dim <= W, "bad patch window");
        p.prow = (H - patch_dim) / patch_stride + 1;
        p.pcol = (W - patch_dim) / patch_stride + 1;
        p.ph = p.pw = patch_dim;
    } else {
        p.prow = p.pcol = 1; p.ph = H; p.pw = W; p.ps = 0;
    }
    p.Ho = (Hv - 1) / 2 + 1;
    p.Wo = (Wv - 1) / 2 + 1;
    p.tiles_y = (p.Ho + TOH - 1) / TOH;
    p.tiles_x = (p.Wo + TOW - 1) / TOW;
    p.Nsamp = (int64_t)B * p.prow * p.pcol;
    p.hwnc = hwnc;
    p.total_tiles = p.Nsamp * p.tiles_y * p.tiles_x;
    constexpr int lds_bytes = (W_TILE + IN_TILE) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)stem_conv7x7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        attr_set = true;
    }
    int64_t grid = p.total_tiles < 4096 ? p.total_tiles : 4096;   // 2 resident per CU x 256 CUs x 8 rounds
    hipLaunchKernelGGL(stem_conv7x7_kernel, dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, p);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_maxpool3x3s2_fwd(const float* in, float* out, int64_t N, int H, int W, int C, int hwnc, void* stream) {
    SSAD_CHECK_ARG(in && out, "null pointer");
    SSAD_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bad shape (C % 4)");
    int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int64_t total4 = N * Ho * Wo * (C / 4);
    SSAD_CHECK_ARG(cdiv64(total4, 256) < (int64_t)2147483647, "too large");
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)cdiv64(total4, 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                       total4, H, W, C / 4, Ho, Wo, N, hwnc);
    SSAD_CHECK_LAUNCH();
    return 0;
}
