// Weight gradient of the stem convolution (7x7 stride 2 pad 3, 3 -> 64) straight from the NCHW image:
//   dW[co][ky][kx][c] = sum_{n,oy,ox} dz[n][oy][ox][co] * img[n][c][2oy+ky-3][2ox+kx-3]      (zero outside the image)
// -- the autograd node of resnet conv1 inside trainer.fit (src/self_supervised/tools.py:270,:303; forward at
// models.py:224).  No im2col buffer: a workgroup stages a 4x32 tile of dz and the 13x69 input pixels under it in LDS and
// contracts over the tile's 128 pixels with v_mfma_f32_32x32x2_f32:  D[co][k] += dz[px][co] * patch[px][k],
// k = (ky*7+kx)*3+c padded 147 -> 160.  The patch operand is pure LDS addressing: lane (r, h) reads
// tin[pixel(h) + koff(r)], and with 213 floats per tile row the 32 k-lanes fall on 32 different banks.
// Workgroups are persistent (grid-stride over tiles), accumulate in registers and leave one [64][160] slab each;
// ssad_wgrad_reduce's kernel adds the slabs in a fixed order (deterministic).
#include "common.h"
#include "ssad.h"
#include <stdlib.h>

namespace {

constexpr int TH = 4, TW = 32;                 // output pixels per tile
constexpr int IH = 2 * TH + 5;                 // 13 input rows
constexpr int IW = 71;                         // >= 2*TW + 5 = 69; 71*3 = 213 = 21 (mod 32): conflict-free k-lanes
constexpr int IN_FLOATS = IH * IW * 3;         // 2769
constexpr int DY_FLOATS = TH * TW * 64;        // 8192
constexpr int KT = 5;                          // 32-wide k tiles (160 >= 147)
constexpr int KPAD = KT * 32;

struct StemWgradParams {
    const float* img;
    const void* dz;       // floats, or halves (TZ = hf)
    float* slab;
    int B, H, W, Hv, Wv, Ho, Wo, tiles_y, tiles_x;
    int64_t total_tiles;
};

template <typename TZ>
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(StemWgradParams p) {
    const TZ* const dzp = (const TZ*)p.dz;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dys = lds;                    // [128 px][64 co]
    float* tin = lds + DY_FLOATS;        // [IH][IW][3]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    int koff[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int k = kt * 32 + r;
        const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
        koff[kt] = k < 147 ? (ky * IW + kx) * 3 + c : 0;        // padded columns read something finite; never stored
    }
    f32x16 acc[2][KT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int tiles_per_sample = p.tiles_y * p.tiles_x;
    const int64_t plane = (int64_t)p.H * p.W;
    const bool direct = p.Hv == p.H && p.Wv == p.W;      // no nearest resize (images of 64 pixels and more): no divisions
    // The operands of a workgroup's NEXT tile are fetched into registers while the current tile's MFMAs run (issued just before
    // the matrix loop, written to LDS after the next barrier)
    constexpr int NDZ = DY_FLOATS / 4 / 256;                 // 8 16-byte pieces of dz per thread
    constexpr int NIN = (IH * IW + 255) / 256;               // 4 input pixels per thread
    f32x4 dzv[NDZ];
    float pv[NIN][3];
    auto fetch_tile = [&](int64_t t) {
        const int64_t n = t / tiles_per_sample;
        const int tt = (int)(t - n * tiles_per_sample);
        const int ty0 = (tt / p.tiles_x) * TH, tx0 = (tt % p.tiles_x) * TW;
        const float* src = p.img + n * 3 * plane;
        // dz tile: TH rows of TW*64 contiguous floats; pixels outside the map contribute zeros
#pragma unroll
        for (int q = 0; q < NDZ; ++q) {
            const int i = tid + 256 * q;
            const int px = i >> 4, c4 = i & 15;
            const int oy = ty0 + (px >> 5), ox = tx0 + (px & 31);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (oy < p.Ho && ox < p.Wo) v = ld4(dzp + ((n * p.Ho + oy) * p.Wo + ox) * 64 + c4 * 4);
            dzv[q] = v;
        }
        // input tile (virtual, i.e. after the nearest resize of models.py:217-219 when the image is below 64x64)
#pragma unroll
        for (int q = 0; q < NIN; ++q) {
            const int i = tid + 256 * q;
            const int iy = i / IW, ix = i - iy * IW;
            const int vy = 2 * ty0 - 3 + iy, vx = 2 * tx0 - 3 + ix;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (i < IH * IW && (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv) {
                const int sy = direct ? vy : (vy * p.H) / p.Hv, sx = direct ? vx : (vx * p.W) / p.Wv;
                const float* s = src + (int64_t)sy * p.W + sx;
                v0 = s[0]; v1 = s[plane]; v2 = s[2 * plane];
            }
            pv[q][0] = v0; pv[q][1] = v1; pv[q][2] = v2;
        }
    };
    if ((int64_t)blockIdx.x < p.total_tiles) fetch_tile(blockIdx.x);
    for (int64_t t = blockIdx.x; t < p.total_tiles; t += gridDim.x) {
        __syncthreads();                 // the previous tile's readers are done
#pragma unroll
        for (int q = 0; q < NDZ; ++q) ((f32x4*)dys)[tid + 256 * q] = dzv[q];
#pragma unroll
        for (int q = 0; q < NIN; ++q) {
            const int i = tid + 256 * q;
            if (i < IH * IW) {
                float* d = tin + i * 3;
                d[0] = pv[q][0]; d[1] = pv[q][1]; d[2] = pv[q][2];
            }
        }
        __syncthreads();
        if (t + gridDim.x < p.total_tiles) fetch_tile(t + gridDim.x);

        // wave w contracts pixel pairs w, w+4, ...: the lane halves hold the pair's two pixels
#pragma unroll 2
        for (int pp = wave; pp < TH * TW / 2; pp += 4) {
            const int px = 2 * pp + h;
            const float a0 = dys[px * 64 + r], a1 = dys[px * 64 + 32 + r];
            const float* bp = tin + (2 * (px >> 5) * IW + 2 * (px & 31)) * 3;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const float b = bp[koff[kt]];
                acc[0][kt] = mfma32(a0, b, acc[0][kt]);
                acc[1][kt] = mfma32(a1, b, acc[1][kt]);
            }
        }
    }

    // ---- the four waves hold partial sums over disjoint pixels: add them through LDS, one 32x32 tile at a time ----
    float* red = lds;                    // [4 waves][32][32]
    float* out = p.slab + (int64_t)blockIdx.x * 64 * KPAD;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[i][kt][e];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = tid + 256 * q, row = idx >> 5, col = idx & 31;
                const float v = ((red[idx] + red[1024 + idx]) + red[2048 + idx]) + red[3072 + idx];
                out[(i * 32 + row) * KPAD + kt * 32 + col] = v;
            }
        }
}

// ---- fp16 OPERANDS (precision-16 step with half tensors): the same tiling on v_mfma_f32_32x32x16_f16 ----
// The fp32 kernel above is bound by its matrix instructions (78.9 GFLOP per batch-256 step on the fp32 MFMAs: 0.93 ms); under the
// reference's pl.Trainer(precision=16) autocast runs conv1 -- forward (stem16.hip) and weight gradient -- on fp16 operands.  dz arrives
// as halves and stays so in LDS ([pixel][co]); the image tile is rounded to halves while it is staged ([row][col][3]).  An MFMA
// contracts over 16 pixels (half a tile row): lane (r, h) supplies, for output channel r / filter column k(r), the 8 pixels
// px0 + 8 h .. + 7 -- eight 2-byte LDS reads at constant offsets per operand (one pixel apart: 128 B for dz, 12 B for the image), which
// the compiler places straight into the register halves (ds_read_u16_d16 / _d16_hi: no packing).  2 x 5 accumulator tiles per wave as
// above; 10 MFMAs per 7 x 8 operand reads.  Same slabs, same fixed-order reduction.
__global__ __launch_bounds__(256, 2) void stem_wgrad16_kernel(StemWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    hf* dys = (hf*)lds;                  // [128 px][64 co]
    hf* tin = dys + DY_FLOATS;           // [IH][IW][3]
    const hf* const dzp = (const hf*)p.dz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    int koff[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int k = kt * 32 + r;
        const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
        koff[kt] = k < 147 ? (ky * IW + kx) * 3 + c : 0;        // padded columns read something finite; never stored
    }
    f32x16 acc[2][KT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int tiles_per_sample = p.tiles_y * p.tiles_x;
    const int64_t plane = (int64_t)p.H * p.W;
    const bool direct = p.Hv == p.H && p.Wv == p.W;
    constexpr int NDZ = DY_FLOATS / 8 / 256;                 // 4 16-byte pieces (8 halves) of dz per thread
    constexpr int NIN = (IH * IW + 255) / 256;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 dzv[NDZ];
    float pv[NIN][3];
    auto fetch_tile = [&](int64_t t) {
        const int64_t n = t / tiles_per_sample;
        const int tt = (int)(t - n * tiles_per_sample);
        const int ty0 = (tt / p.tiles_x) * TH, tx0 = (tt % p.tiles_x) * TW;
        const float* src = p.img + n * 3 * plane;
#pragma unroll
        for (int q = 0; q < NDZ; ++q) {
            const int i = tid + 256 * q;
            const int px = i >> 3, c8 = i & 7;
            const int oy = ty0 + (px >> 5), ox = tx0 + (px & 31);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (oy < p.Ho && ox < p.Wo) v = *(const u32x4*)(dzp + ((n * p.Ho + oy) * p.Wo + ox) * 64 + c8 * 8);
            dzv[q] = v;
        }
#pragma unroll
        for (int q = 0; q < NIN; ++q) {
            const int i = tid + 256 * q;
            const int iy = i / IW, ix = i - iy * IW;
            const int vy = 2 * ty0 - 3 + iy, vx = 2 * tx0 - 3 + ix;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (i < IH * IW && (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv) {
                const int sy = direct ? vy : (vy * p.H) / p.Hv, sx = direct ? vx : (vx * p.W) / p.Wv;
                const float* sp = src + (int64_t)sy * p.W + sx;
                v0 = sp[0]; v1 = sp[plane]; v2 = sp[2 * plane];
            }
            pv[q][0] = v0; pv[q][1] = v1; pv[q][2] = v2;
        }
    };
    if ((int64_t)blockIdx.x < p.total_tiles) fetch_tile(blockIdx.x);
    for (int64_t t = blockIdx.x; t < p.total_tiles; t += gridDim.x) {
        __syncthreads();                 // the previous tile's readers are done
#pragma unroll
        for (int q = 0; q < NDZ; ++q) ((u32x4*)dys)[tid + 256 * q] = dzv[q];
#pragma unroll
        for (int q = 0; q < NIN; ++q) {
            const int i = tid + 256 * q;
            if (i < IH * IW) {
                hf* d = tin + i * 3;
                d[0] = (hf)pv[q][0]; d[1] = (hf)pv[q][1]; d[2] = (hf)pv[q][2];
            }
        }
        __syncthreads();
        if (t + gridDim.x < p.total_tiles) fetch_tile(t + gridDim.x);

        // wave w contracts pixel groups w, w + 4 (16 pixels = half a tile row each): lane half h takes pixels px0 + 8 h .. + 7
#pragma unroll
        for (int g = wave; g < TH * TW / 16; g += 4) {
            const int px0 = g * 16 + 8 * h;
            const hf* ap = dys + px0 * 64 + r;
            const hf* bp = tin + (2 * (px0 >> 5) * IW + 2 * (px0 & 31)) * 3;
            f16x8 a0, a1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { a0[j] = ap[j * 64]; a1[j] = ap[j * 64 + 32]; }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                f16x8 b;
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = bp[koff[kt] + j * 6];
                acc[0][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[0][kt], 0, 0, 0);
                acc[1][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b, acc[1][kt], 0, 0, 0);
            }
        }
    }

    // ---- the four waves hold partial sums over disjoint pixels: add them through LDS, one 32x32 tile at a time ----
    float* red = lds;                    // [4 waves][32][32]
    float* out = p.slab + (int64_t)blockIdx.x * 64 * KPAD;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[i][kt][e];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = tid + 256 * q, row = idx >> 5, col = idx & 31;
                const float v = ((red[idx] + red[1024 + idx]) + red[2048 + idx]) + red[3072 + idx];
                out[(i * 32 + row) * KPAD + kt * 32 + col] = v;
            }
        }
}

int stem_wgrad_blocks(int64_t total_tiles) {
    static const int target = getenv("SSAD_STEM_WGRAD_BLOCKS") ? atoi(getenv("SSAD_STEM_WGRAD_BLOCKS")) : 512;
    return (int)(total_tiles < target ? total_tiles : target);
}

}  // namespace

extern "C" int64_t ssad_stem_wgrad_workspace(int B, int H, int W) {
    const int Hv = (H < 64 || W < 64) ? 64 : H, Wv = (H < 64 || W < 64) ? 64 : W;
    const int Ho = (Hv - 1) / 2 + 1, Wo = (Wv - 1) / 2 + 1;
    const int64_t tiles = (int64_t)B * ((Ho + TH - 1) / TH) * ((Wo + TW - 1) / TW);
    return (int64_t)stem_wgrad_blocks(tiles) * 64 * KPAD;
}

static int stem_wgrad_impl(const float* img, const void* dz, int dz_half, float* dw, int B, int H, int W, int64_t dz_elems, int to_oihw,
                           int accumulate, float* workspace, void* stream) {
    SSAD_CHECK_ARG(img && dz && dw && workspace, "null pointer");
    SSAD_CHECK_ARG(B > 0 && H > 0 && W > 0, "empty shape");
    StemWgradParams p;
    p.img = img; p.dz = dz; p.slab = workspace;
    p.B = B; p.H = H; p.W = W;
    p.Hv = (H < 64 || W < 64) ? 64 : H;
    p.Wv = (H < 64 || W < 64) ? 64 : W;
    p.Ho = (p.Hv - 1) / 2 + 1;
    p.Wo = (p.Wv - 1) / 2 + 1;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.total_tiles = (int64_t)B * p.tiles_y * p.tiles_x;
    // dz is read over B x Ho x Wo x 64 as derived from the IMAGE's extents (round 4: a half-size dz from a stem variant that skipped the
    // nearest resize was read out of bounds here): the caller states what its buffer holds
    SSAD_CHECK_ARG(dz_elems == (int64_t)B * p.Ho * p.Wo * 64, "dz does not hold B x Ho x Wo x 64 elements for these images");
    const int nblk = stem_wgrad_blocks(p.total_tiles);
    // dz_half: 1 = halves read into the fp32-MFMA kernel (exact products of the stored values); 2 = fp16 operands (image rounded)
    if (dz_half == 2) hipLaunchKernelGGL(stem_wgrad16_kernel, dim3(nblk), dim3(256), (DY_FLOATS + IN_FLOATS) * 4, (hipStream_t)stream, p);
    else if (dz_half) hipLaunchKernelGGL(stem_wgrad_kernel<hf>, dim3(nblk), dim3(256), (DY_FLOATS + IN_FLOATS) * 4, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(stem_wgrad_kernel<float>, dim3(nblk), dim3(256), (DY_FLOATS + IN_FLOATS) * 4, (hipStream_t)stream, p);
    SSAD_CHECK_LAUNCH();
    return ssad_wgrad_reduce(workspace, dw, nblk, 64, KPAD, 7, 7, 3, to_oihw, accumulate, stream);
}

extern "C" int ssad_stem_wgrad(const float* img, const float* dz, float* dw, int B, int H, int W, int64_t dz_elems, int to_oihw,
                               int accumulate, float* workspace, void* stream) {
    return stem_wgrad_impl(img, dz, 0, dw, B, H, W, dz_elems, to_oihw, accumulate, workspace, stream);
}

// dz stored as halves (precision-16 step); products and sums in fp32 as above
extern "C" int ssad_stem_wgrad_h(const float* img, const void* dz, float* dw, int B, int H, int W, int64_t dz_elems, int to_oihw,
                                 int accumulate, float* workspace, void* stream) {
    static const int f16 = getenv("SSAD_STEM_WGRAD16") ? atoi(getenv("SSAD_STEM_WGRAD16")) : 1;
    return stem_wgrad_impl(img, dz, f16 ? 2 : 1, dw, B, H, W, dz_elems, to_oihw, accumulate, workspace, stream);
}
