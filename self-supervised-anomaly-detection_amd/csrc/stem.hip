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
#include <type_traits>

namespace {

constexpr int STEM_MAX_GRID = 4096;   // persistent workgroups of the stem conv = rows of its statistics partials (ssad_stem_stats_rows)


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
    double* stats;        // optional [gridDim.x][2][64]: per-workgroup sums / sums of squares of the raw outputs it wrote
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

    double st0[2] = {0.0, 0.0}, st1[2] = {0.0, 0.0};      // train-mode BatchNorm statistics of this lane's channels
    const int tiles_per_sample = p.tiles_y * p.tiles_x;
    const bool direct = p.ph == p.Hv && p.pw == p.Wv;      // no nearest resize (windows of 64 pixels and more): no divisions
    const int64_t plane = (int64_t)p.H * p.W;
    // The (virtual, zero-padded) input tile of a workgroup's NEXT tile is fetched into registers while the current tile's MFMAs
    // run (the loads are issued just before the matrix loop and land under it; they reach LDS after the next barrier)
    constexpr int NIN = (TIH * TIW + 255) / 256;             // 6 pixels per thread
    float pv[NIN][3];
    auto fetch_tile = [&](int64_t t) {
        const int64_t n = t / tiles_per_sample;
        const int tt = (int)(t - n * tiles_per_sample);
        const int ty0 = (tt / p.tiles_x) * TOH, tx0 = (tt % p.tiles_x) * TOW;
        const int P = p.prow * p.pcol;
        const int b = (int)(n / P);
        const int pi = (int)(n - (int64_t)b * P);
        const int y0 = (pi / p.pcol) * p.ps, x0 = (pi % p.pcol) * p.ps;
        const float* src = p.img + (int64_t)b * 3 * plane;
#pragma unroll
        for (int q = 0; q < NIN; ++q) {
            const int i = tid + 256 * q;
            const int iy = i / TIW, ix = i - iy * TIW;
            const int vy = 2 * ty0 - 3 + iy, vx = 2 * tx0 - 3 + ix;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (i < TIH * TIW && (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv) {
                const int sy = y0 + (direct ? vy : (vy * p.ph) / p.Hv), sx = x0 + (direct ? vx : (vx * p.pw) / p.Wv);
                const float* s = src + (int64_t)sy * p.W + sx;
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
                float* d = tin + i * 3;
                d[0] = pv[q][0]; d[1] = pv[q][1]; d[2] = pv[q][2];
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
        // Whole tiles (all 32 columns exist): straight-line code -- a uniform base per output row, a constant per-lane byte offset, the
        // register's pixel as a scalar offset; statistics as packed fp32 pairs (one instruction per two values).  Every VALU
        // instruction here is time the wave's matrix pipe stands still (tools/micro/conv32w_trace.hip, round 6); the general form
        // below spent ~10 per value on bounds tests, 64-bit pixel arithmetic and flag branches.
        if (tx0 + 32 <= p.Wo) {
            const int64_t pst = p.hwnc ? p.Nsamp * 256 : 256;                      // bytes between neighbouring pixels of a row
            const unsigned lane_off = (unsigned)((int64_t)(4 * h) * pst + r * 4);
            typedef float f32x2v __attribute__((ext_vector_type(2)));
            auto rows = [&](auto stats_tag) __attribute__((always_inline)) {
                constexpr bool ST = decltype(stats_tag)::value;
                const float lowb = p.relu ? 0.f : -__builtin_huge_valf();
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int oy = ty0 + 2 * wave + i;
                    if (oy >= p.Ho) continue;
                    const int64_t pix0 = p.hwnc ? ((int64_t)oy * p.Wo + tx0) * p.Nsamp + n : (n * p.Ho + oy) * p.Wo + tx0;
                    char* const ob = (char*)(p.out + pix0 * 64);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x2v ps = {0.f, 0.f}, pq = {0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < 16; e += 2) {
                            f32x2v v = {acc[i][j][e], acc[i][j][e + 1]};
                            if (ST) {
                                ps += v;
                                pq = v * v + pq;
                            } else {
                                v = v * f32x2v{sc[j], sc[j]} + f32x2v{sh[j], sh[j]};
                                v[0] = fmaxf(v[0], lowb); v[1] = fmaxf(v[1], lowb);
                            }
#pragma unroll
                            for (int k = 0; k < 2; ++k)
                                *(float*)(ob + (int64_t)(((e + k) & 3) + 8 * ((e + k) >> 2)) * pst + j * 128 + lane_off) = v[k];
                        }
                        if (ST) { st0[j] += (double)ps[0] + (double)ps[1]; st1[j] += (double)pq[0] + (double)pq[1]; }
                    }
                }
            };
            if (p.stats && !p.scale && !p.shift && !p.relu) { rows(std::true_type()); continue; }
            if (!p.stats) { rows(std::false_type()); continue; }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = ty0 + 2 * wave + i;
            if (oy >= p.Ho) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // statistics: the 16 values of this lane's channel in float, then one double add per tile row (the conv epilogues of
                // conv_igemm.hip sum 128 rows in float the same way); 64 double-precision FMAs per tile row cost the wave ~10 % of it
                float fs = 0.f, fq = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int ox = tx0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (ox < p.Wo) {
                        if (p.stats) { fs += acc[i][j][e]; fq += acc[i][j][e] * acc[i][j][e]; }
                        float v = acc[i][j][e] * sc[j] + sh[j];
                        if (p.relu) v = fmaxf(v, 0.f);
                        const int64_t pix = p.hwnc ? ((int64_t)oy * p.Wo + ox) * p.Nsamp + n : (n * p.Ho + oy) * p.Wo + ox;
                        p.out[pix * 64 + j * 32 + r] = v;
                    }
                }
                if (p.stats) { st0[j] += (double)fs; st1[j] += (double)fq; }
            }
        }
    }
    if (p.stats) {
        // lane halves -> one value per channel per wave, the four waves in a fixed order through LDS (the weights are dead)
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
}

// ---------------------------------------------------------------------------------------------
// Patch-scoring stem: 32x32 window, exact 2x nearest upsample, conv7x7/2 + affine + ReLU + max-pool 3x3/2, fused.
//
// nearest-2x followed by a stride-2 7x7 conv touches each SOURCE pixel through a fixed group of taps:
//   ky -> source row oy-2+a with a = 0:{0} 1:{1,2} 2:{3,4} 3:{5,6}   (same for kx -> b), and a virtual row is in
// the zero padding exactly when its source row is outside [0,32).  So
//   out[oy][ox][co] = sum_{a,b<4,c} src[oy-2+a][ox-2+b][c] * Wf[co][a][b][c],  Wf = group sums of W   (K 147 -> 48)
// -- the same real number as the reference's sum, rounded differently (1 ulp-level).  The 32x32x64 conv map goes
// through an LDS ring of 9 rows (32 channels at a time), is pooled there and only the 16x16x64 result is stored.
// ---------------------------------------------------------------------------------------------
constexpr int SP_SW = 40, SP_SH = 36;                 // source tile: rows -2..33, cols -2..37 (zero halo)
constexpr int SP_SRC = SP_SH * SP_SW * 3;             // 4320 floats
constexpr int SP_WF = 24 * 2 * 64;                    // folded weights [a*2+q][c][h][co] = 3072 floats
constexpr int SP_CB = 9 * 32 * 32;                    // conv-row ring [slot][ox][32 ch] = 9216 floats

struct StemPatchParams {
    const float* img;
    const float* wf;
    const float* scale;
    const float* shift;
    float* out;
    int B, H, W, ps, prow, pcol, hwnc;
    int skip_lo, skip_hi;        // pooled positions skip_lo <= py, px <= skip_hi are neither pooled nor stored (skip_lo > skip_hi: none)
    int64_t Nsamp;
};

__global__ __launch_bounds__(256, 2) void stem_patch_fused_kernel(StemPatchParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* src = lds + SP_WF;
    float* cb = src + SP_SRC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (int i = tid; i < SP_WF / 4; i += 256) ((f32x4*)wl)[i] = ((const f32x4*)p.wf)[i];
    const int P = p.prow * p.pcol;
    const int64_t plane = (int64_t)p.H * p.W;

    // the source window of a workgroup's NEXT patch is fetched into registers while the current patch is computed
    constexpr int NSRC = (SP_SH * SP_SW + 255) / 256;        // 6 pixels per thread
    float pv[NSRC][3];
    auto fetch_patch = [&](int64_t n) {
        const int b = (int)(n / P);
        const int pi = (int)(n - (int64_t)b * P);
        const int y0 = (pi / p.pcol) * p.ps, x0 = (pi % p.pcol) * p.ps;
        const float* im = p.img + (int64_t)b * 3 * plane;
#pragma unroll
        for (int q = 0; q < NSRC; ++q) {
            const int i = tid + 256 * q;
            const int ty = i / SP_SW, tx = i - ty * SP_SW;
            const int sy = ty - 2, sx = tx - 2;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (i < SP_SH * SP_SW && (unsigned)sy < 32u && (unsigned)sx < 32u) {
                const float* s = im + (int64_t)(y0 + sy) * p.W + x0 + sx;
                v0 = s[0]; v1 = s[plane]; v2 = s[2 * plane];
            }
            pv[q][0] = v0; pv[q][1] = v1; pv[q][2] = v2;
        }
    };
    if ((int64_t)blockIdx.x < p.Nsamp) fetch_patch(blockIdx.x);
    for (int64_t n = blockIdx.x; n < p.Nsamp; n += gridDim.x) {
        __syncthreads();                                    // previous patch fully consumed
#pragma unroll
        for (int q = 0; q < NSRC; ++q) {
            const int i = tid + 256 * q;
            if (i < SP_SH * SP_SW) {
                float* d = src + i * 3;
                d[0] = pv[q][0]; d[1] = pv[q][1]; d[2] = pv[q][2];
            }
        }
        if (n + gridDim.x < p.Nsamp) fetch_patch(n + gridDim.x);
        for (int pass = 0; pass < 2; ++pass) {
            const float sc = p.scale ? p.scale[pass * 32 + r] : 1.f;
            const float sh = p.shift ? p.shift[pass * 32 + r] : 0.f;
            for (int i = tid; i < 32 * 32; i += 256) cb[i] = 0.f;      // ring slot 0 = conv row -1 (below every ReLU output)
            __syncthreads();
            for (int t = 0; t < 4; ++t) {
                f32x16 acc0, acc1;
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
                const int oy = 8 * t + 2 * wave;
                const float* a0 = src + ((oy + 0) * SP_SW + r + h) * 3;
                const float* a1 = src + ((oy + 1) * SP_SW + r + h) * 3;
                const float* bw = wl + h * 64 + pass * 32 + r;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const int ko = (a * SP_SW + 2 * q) * 3 + c;
                            const float w = bw[((a * 2 + q) * 3 + c) * 128];
                            acc0 = mfma32(a0[ko], w, acc0);
                            acc1 = mfma32(a1[ko], w, acc1);
                        }
                float* c0 = cb + ((1 + 2 * wave) * 32) * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ox = (e & 3) + 8 * (e >> 2) + 4 * h;
                    c0[ox * 32] = fmaxf(acc0[e] * sc + sh, 0.f);
                    c0[(32 + ox) * 32] = fmaxf(acc1[e] * sc + sh, 0.f);
                }
                __syncthreads();
                // pool: pooled rows 4t .. 4t+3 from ring slots 2j .. 2j+2; 512 float4 items
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int item = tid + 256 * k;
                    const int c4 = item & 7, px = (item >> 3) & 15, j = item >> 7;
                    if (4 * t + j >= p.skip_lo && 4 * t + j <= p.skip_hi && px >= p.skip_lo && px <= p.skip_hi) continue;
                    f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = -1; dx < 2; ++dx) {
                            const int cx = 2 * px + dx;
                            if (cx < 0) continue;
                            const f32x4 v = *(const f32x4*)(cb + ((2 * j + dy) * 32 + cx) * 32 + c4 * 4);
                            m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
                        }
                    const int py = 4 * t + j;
                    const int64_t pix = p.hwnc ? ((int64_t)py * 16 + px) * p.Nsamp + n : (n * 16 + py) * 16 + px;
                    *(f32x4*)(p.out + pix * 64 + pass * 32 + c4 * 4) = m;
                }
                __syncthreads();
                ((f32x4*)cb)[tid] = ((const f32x4*)(cb + 8 * 32 * 32))[tid];      // slot 8 -> slot 0 for the next tile
                __syncthreads();
            }
        }
    }
}

// The BORDER of the same pooled map only (round 6, layer1 shared between overlapping patches): pooled rows / columns 0, 1 and 15 of
// every patch -- the positions that differ from the per-image pooled map because the patch's own zero padding is within reach (stem rows
// 0, 1, 31) -- 87 of the 256 positions.  Everything else the first ring conv reads (rows / columns 2, 3, 13, 14) is copied from the dense
// map (ssad_patch_gather_hwnc_band).  The conv values needed are rows 0-3 and 29-31 (7 tiles of 32 pixels) and columns 0-3 and 29-31 of
// rows 3-29 (189 pixels = 6 tiles): 13 tiles instead of 32, each the same 24 MFMAs in the same order as the fused kernel above -- the
// values are bit-identical to its output at these positions.  Output position-major [16][16][Nsamp][64] only.
constexpr int SB_CB = 7 * 32 * 32;                    // conv buffer: 7 rows x 32 pixels (rows phase) / 189 pixels (columns phase) x 32 channels

__global__ __launch_bounds__(256, 2) void stem_patch_border_kernel(StemPatchParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* src = lds + SP_WF;
    float* cb = src + SP_SRC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (int i = tid; i < SP_WF / 4; i += 256) ((f32x4*)wl)[i] = ((const f32x4*)p.wf)[i];
    const int P = p.prow * p.pcol;
    const int64_t plane = (int64_t)p.H * p.W;
    constexpr int NSRC = (SP_SH * SP_SW + 255) / 256;
    float pv[NSRC][3];
    auto fetch_patch = [&](int64_t n) {
        const int b = (int)(n / P);
        const int pi = (int)(n - (int64_t)b * P);
        const int y0 = (pi / p.pcol) * p.ps, x0 = (pi % p.pcol) * p.ps;
        const float* im = p.img + (int64_t)b * 3 * plane;
#pragma unroll
        for (int q = 0; q < NSRC; ++q) {
            const int i = tid + 256 * q;
            const int ty = i / SP_SW, tx = i - ty * SP_SW;
            const int sy = ty - 2, sx = tx - 2;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (i < SP_SH * SP_SW && (unsigned)sy < 32u && (unsigned)sx < 32u) {
                const float* s = im + (int64_t)(y0 + sy) * p.W + x0 + sx;
                v0 = s[0]; v1 = s[plane]; v2 = s[2 * plane];
            }
            pv[q][0] = v0; pv[q][1] = v1; pv[q][2] = v2;
        }
    };
    // this lane's pixels (A rows) of the wave's two tiles in each phase: rows phase tiles 2 wave, 2 wave + 1 of the row list
    // {0, 1, 2, 3, 29, 30, 31}; columns phase tiles 2 wave, 2 wave + 1 of the pixel list q = (oy - 3) * 7 + ci, ci -> ox {0, 1, 2, 3, 29, 30, 31}
    auto row_of = [](int t) { return t < 4 ? t : 25 + t; };                    // t = 4, 5, 6 -> 29, 30, 31
    const int ta0 = 2 * wave, ta1 = 2 * wave + 1 < 7 ? 2 * wave + 1 : 2 * wave;  // (wave 3: one tile, computed twice, stored once)
    const float* pa0 = src + (row_of(ta0) * SP_SW + r + h) * 3;
    const float* pa1 = src + (row_of(ta1) * SP_SW + r + h) * 3;
    const float *pb0, *pb1;
    {
        int q0 = (2 * wave) * 32 + r, q1 = (2 * wave + 1) * 32 + r;
        q0 = q0 < 189 ? q0 : 188; q1 = q1 < 189 ? q1 : 188;
        const int c0 = q0 % 7, c1 = q1 % 7;
        pb0 = src + ((3 + q0 / 7) * SP_SW + (c0 < 4 ? c0 : 25 + c0) + h) * 3;
        pb1 = src + ((3 + q1 / 7) * SP_SW + (c1 < 4 ? c1 : 25 + c1) + h) * 3;
    }
    auto two_tiles = [&](const float* a0, const float* a1, int pass, f32x16& acc0, f32x16& acc1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        const float* bw = wl + h * 64 + pass * 32 + r;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int ko = (a * SP_SW + 2 * q) * 3 + c;
                    const float w = bw[((a * 2 + q) * 3 + c) * 128];
                    acc0 = mfma32(a0[ko], w, acc0);
                    acc1 = mfma32(a1[ko], w, acc1);
                }
    };
    if ((int64_t)blockIdx.x < p.Nsamp) fetch_patch(blockIdx.x);
    for (int64_t n = blockIdx.x; n < p.Nsamp; n += gridDim.x) {
        __syncthreads();                                    // previous patch fully consumed
#pragma unroll
        for (int q = 0; q < NSRC; ++q) {
            const int i = tid + 256 * q;
            if (i < SP_SH * SP_SW) {
                float* d = src + i * 3;
                d[0] = pv[q][0]; d[1] = pv[q][1]; d[2] = pv[q][2];
            }
        }
        if (n + gridDim.x < p.Nsamp) fetch_patch(n + gridDim.x);
        __syncthreads();
        for (int pass = 0; pass < 2; ++pass) {
            const float sc = p.scale ? p.scale[pass * 32 + r] : 1.f;
            const float sh = p.shift ? p.shift[pass * 32 + r] : 0.f;
            f32x16 acc0, acc1;
            // ---- rows phase: conv rows 0-3, 29-31 -> cb[t][ox][ch] ----
            two_tiles(pa0, pa1, pass, acc0, acc1);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ox = (e & 3) + 8 * (e >> 2) + 4 * h;
                cb[((2 * wave) * 32 + ox) * 32 + r] = fmaxf(acc0[e] * sc + sh, 0.f);
                if (2 * wave + 1 < 7) cb[((2 * wave + 1) * 32 + ox) * 32 + r] = fmaxf(acc1[e] * sc + sh, 0.f);
            }
            __syncthreads();
            for (int item = tid; item < 3 * 16 * 8; item += 256) {      // pooled rows 0, 1, 15: all 16 columns
                const int c4 = item & 7, px = (item >> 3) & 15, pr = item >> 7;
                const int t_lo = pr == 0 ? 0 : pr == 1 ? 1 : 4, t_hi = pr == 0 ? 1 : pr == 1 ? 3 : 6;
                f32x4 m = {0.f, 0.f, 0.f, 0.f};
                for (int t = t_lo; t <= t_hi; ++t)
#pragma unroll
                    for (int dx = -1; dx < 2; ++dx) {
                        const int cx = 2 * px + dx;
                        if (cx < 0) continue;
                        const f32x4 v = *(const f32x4*)(cb + (t * 32 + cx) * 32 + c4 * 4);
                        m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
                    }
                const int py = pr < 2 ? pr : 15;
                *(f32x4*)(p.out + (((int64_t)py * 16 + px) * p.Nsamp + n) * 64 + pass * 32 + c4 * 4) = m;
            }
            __syncthreads();
            // ---- columns phase: conv columns 0-3, 29-31 of rows 3-29 -> cb[q][ch] ----
            if (wave < 3) {
                two_tiles(pb0, pb1, pass, acc0, acc1);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int q0 = (2 * wave) * 32 + i, q1 = q0 + 32;
                    if (q0 < 189) cb[q0 * 32 + r] = fmaxf(acc0[e] * sc + sh, 0.f);
                    if (q1 < 189) cb[q1 * 32 + r] = fmaxf(acc1[e] * sc + sh, 0.f);
                }
            }
            __syncthreads();
            for (int item = tid; item < 13 * 3 * 8; item += 256) {      // pooled rows 2-14, columns 0, 1, 15
                const int c4 = item & 7, k = item >> 3, pk = k % 3, py = 2 + k / 3;
                const int ci_lo = pk == 0 ? 0 : pk == 1 ? 1 : 4, ci_hi = pk == 0 ? 1 : pk == 1 ? 3 : 6;
                f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dy = -1; dy < 2; ++dy)
                    for (int ci = ci_lo; ci <= ci_hi; ++ci) {
                        const f32x4 v = *(const f32x4*)(cb + ((2 * py + dy - 3) * 7 + ci) * 32 + c4 * 4);
                        m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
                    }
                const int px = pk < 2 ? pk : 15;
                *(f32x4*)(p.out + (((int64_t)py * 16 + px) * p.Nsamp + n) * 64 + pass * 32 + c4 * 4) = m;
            }
            __syncthreads();
        }
    }
}

// OIHW [64][3][7][7] -> folded [ (a*2+q)*3 + c ][h][co], b = 2q+h, a/b tap groups {0},{1,2},{3,4},{5,6}
__global__ void pack_stem_weight_folded_kernel(const float* __restrict__ w, float* __restrict__ wf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SP_WF) return;
    const int co = i & 63, hh = (i >> 6) & 1, s = i >> 7;
    const int c = s % 3, q = (s / 3) & 1, a = s / 6;
    const int b = 2 * q + hh;
    const int ky0 = a == 0 ? 0 : 2 * a - 1, ky1 = a == 0 ? 0 : 2 * a;
    const int kx0 = b == 0 ? 0 : 2 * b - 1, kx1 = b == 0 ? 0 : 2 * b;
    float acc = 0.f;
    for (int ky = ky0; ky <= ky1; ++ky)
        for (int kx = kx0; kx <= kx1; ++kx) acc += w[((co * 3 + c) * 7 + ky) * 7 + kx];
    wf[i] = acc;
}

// ohwi: the source is [64][7][7][3] (the layout of the training step's parameter arena) instead of OIHW [64][3][7][7]
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, float* __restrict__ wk, int ohwi = 0) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;     // over 168*64
    if (i >= 168 * 64) return;
    int co = i & 63, kl = i >> 6;
    int hh = kl & 1, rest = kl >> 1;
    int c = rest % 3, q = (rest / 3) & 3, ky = rest / 12;
    int kx = 2 * q + hh;
    const int src = ohwi ? ((co * 7 + ky) * 7 + (kx < 7 ? kx : 0)) * 3 + c : ((co * 3 + c) * 7 + ky) * 7 + (kx < 7 ? kx : 0);
    const float v = w[src];
    wk[i] = kx < 7 ? v : 0.f;
}

// NHWC 3x3 stride-2 pad-1 max-pool, 4 channels per thread.
// With `bn` (mean, invstd, gamma, beta of a train-mode BatchNorm) the input is the raw convolution z and every tap is
// first mapped through relu((z - mean) * invstd * gamma + beta) -- the expression of bn_apply_fwd -- so the activation
// between conv1/bn1/relu and the pool never exists in HBM.
struct PoolBN { const float* mean; const float* invstd; const float* gamma; const float* beta; };

// A lane owns E = 4 (float) / 8 (half) consecutive channels: 16-byte accesses either way (totalE, CE in units of E).
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ in, T* __restrict__ out, uint8_t* __restrict__ idx,
                                    int64_t totalE, int H, int W, int CE, int Ho, int Wo, int64_t N, int hwnc, PoolBN bn,
                                    T* __restrict__ zwin = nullptr) {
    constexpr int E = Lane<T>::E;
    using V = typename Lane<T>::vec;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= totalE) return;
    int cq = (int)(i % CE);
    int64_t pix = i / CE;
    int ox, oy;
    int64_t n;
    if (hwnc) {                       // i enumerates [Ho][Wo][N][CE]
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
    V m = -INFINITY, zw = 0.f;                // zw: the RAW input value of the winner (zwin: what the BatchNorm backward reduction needs of z)
    int am[E];                                // window slot (dy*3+dx) of the FIRST maximum, PyTorch's tie rule
#pragma unroll
    for (int k = 0; k < E; ++k) am[k] = 0;
    V mu = 0.f, is = 0.f, ga = 0.f, be = 0.f;
    if (bn.mean) {
        mu = ldpar<V>(bn.mean, cq); is = ldpar<V>(bn.invstd, cq);
        ga = ldpar<V>(bn.gamma, cq); be = ldpar<V>(bn.beta, cq);
    }
    // All nine taps are requested before any is used (branch-free: a tap outside the map loads the window's centre, which always
    // exists, and is skipped below).  With the loads inside the bounds branches hipcc waited for each one before the next --
    // nine memory round trips per thread: 318 us for 0.87 GB over halves (round 6, ISA inspection).
    V tap[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int y = oy * 2 - 1 + dy, x = ox * 2 - 1 + dx;
            const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const int yc = ok ? y : oy * 2, xc = ok ? x : ox * 2;
            const int64_t ip = hwnc ? ((int64_t)yc * W + xc) * N + n : (n * H + yc) * W + xc;
            tap[dy * 3 + dx] = ldv(in + E * (ip * CE + cq));
        }
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        int y = oy * 2 - 1 + dy;
        if ((unsigned)y >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            int x = ox * 2 - 1 + dx;
            if ((unsigned)x >= (unsigned)W) continue;
            V v = tap[dy * 3 + dx];
            const V raw = v;
            if (bn.mean) {
                // (half tensors: the BatchNorm output is itself a stored half under autocast, so candidates are compared -- and tie --
                // as halves)
#pragma unroll
                for (int k = 0; k < E; ++k) v[k] = stored<T>(fmaxf((v[k] - mu[k]) * is[k] * ga[k] + be[k], 0.f));
            }
#pragma unroll
            for (int k = 0; k < E; ++k)
                if (v[k] > m[k]) { m[k] = v[k]; am[k] = dy * 3 + dx; zw[k] = raw[k]; }
        }
    }
    stv(out + E * i, m);
    if (zwin) stv(zwin + E * i, zw);
    if (idx) {
#pragma unroll
        for (int q = 0; q < E / 4; ++q)
            ((uint32_t*)idx)[i * (E / 4) + q] = (uint32_t)am[4 * q] | ((uint32_t)am[4 * q + 1] << 8) | ((uint32_t)am[4 * q + 2] << 16) |
                                                ((uint32_t)am[4 * q + 3] << 24);
    }
}

}  // namespace

extern "C" int ssad_pack_stem_weight(const float* w_oihw, float* wk, void* stream) {
    SSAD_CHECK_ARG(w_oihw && wk, "null pointer");
    hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((168 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, wk, 0);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// The same pack from an OHWI filter [64][7][7][3] -- how the training step's parameter arena holds conv1 -- so that the step needs no
// OIHW copy of the weight in front of it (round 6: one ~5 us copy launch per step less).
extern "C" int ssad_pack_stem_weight_ohwi(const float* w_ohwi, float* wk, void* stream) {
    SSAD_CHECK_ARG(w_ohwi && wk, "null pointer");
    hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((168 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_ohwi, wk, 1);
    SSAD_CHECK_LAUNCH();
    return 0;
}

static int stem_fwd_impl(const float* img, int B, int H, int W, int patch_dim, int patch_stride, int Hv, int Wv,
                         const float* wk, const float* scale, const float* shift, int relu, int hwnc, float* out,
                         void* stream, double* stats, int* stat_rows) {
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
    p.stats = stats;
    p.total_tiles = p.Nsamp * p.tiles_y * p.tiles_x;
    constexpr int lds_bytes = (W_TILE + IN_TILE) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        SSAD_SET_DYN_LDS(stem_conv7x7_kernel, lds_bytes);
        attr_set = true;
    }
    int64_t grid = p.total_tiles < STEM_MAX_GRID ? p.total_tiles : STEM_MAX_GRID;   // 2 resident per CU x 256 CUs x 8 rounds
    hipLaunchKernelGGL(stem_conv7x7_kernel, dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, p);
    SSAD_CHECK_LAUNCH();
    if (stat_rows) *stat_rows = (int)grid;
    return 0;
}

extern "C" int ssad_stem_fwd(const float* img, int B, int H, int W, int patch_dim, int patch_stride, int Hv, int Wv,
                             const float* wk, const float* scale, const float* shift, int relu, int hwnc, float* out,
                             void* stream) {
    return stem_fwd_impl(img, B, H, W, patch_dim, patch_stride, Hv, Wv, wk, scale, shift, relu, hwnc, out, stream, nullptr, nullptr);
}

// conv1 of the stem in training: the raw convolution z (NHWC) AND the train-mode BatchNorm statistics of bn1 in one pass
// (resnet conv1 + bn1 under trainer.fit, models.py:224): per-workgroup fp64 partial sums from the accumulators, finalised as
// ssad_bn_stats does.  workspace: ssad_stem_stats_rows() * 128 doubles.
extern "C" int ssad_stem_stats_rows(void) { return STEM_MAX_GRID; }

extern "C" int ssad_stem_fwd_stats(const float* img, int B, int H, int W, int Hv, int Wv, const float* wk, float* out, float eps,
                                   float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                                   double* workspace, void* stream) {
    SSAD_CHECK_ARG(mean && invstd && workspace, "null pointer");
    int rows = 0;
    const int rc = stem_fwd_impl(img, B, H, W, 0, 0, Hv, Wv, wk, nullptr, nullptr, 0, 0, out, stream, workspace, &rows);
    if (rc) return rc;
    const int Ho = (Hv - 1) / 2 + 1, Wo = (Wv - 1) / 2 + 1;
    return ssad_bn_finalize_partials(workspace, rows, (int64_t)B * Ho * Wo, 64, eps, momentum, mean, invstd, running_mean,
                                     running_var, stream);
}

extern "C" int ssad_pack_stem_weight_folded(const float* w_oihw, float* wf, void* stream) {
    SSAD_CHECK_ARG(w_oihw && wf, "null pointer");
    hipLaunchKernelGGL(pack_stem_weight_folded_kernel, dim3((SP_WF + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, wf);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// ... leaving out the pooled positions skip_lo <= py, px <= skip_hi of every patch (the patch-scoring pass with layer1 shared between
// overlapping patches: its first ring conv reads the pooled map within one position of the outputs it computes)
extern "C" int ssad_stem_patch_pool_fwd_ring(const float* img, int B, int H, int W, int patch_stride, const float* wf,
                                             const float* scale, const float* shift, int hwnc, int skip_lo, int skip_hi, float* out,
                                             void* stream);
extern "C" int ssad_stem_patch_pool_fwd(const float* img, int B, int H, int W, int patch_stride, const float* wf,
                                        const float* scale, const float* shift, int hwnc, float* out, void* stream) {
    return ssad_stem_patch_pool_fwd_ring(img, B, H, W, patch_stride, wf, scale, shift, hwnc, 1, 0, out, stream);
}
extern "C" int ssad_stem_patch_pool_fwd_ring(const float* img, int B, int H, int W, int patch_stride, const float* wf,
                                             const float* scale, const float* shift, int hwnc, int skip_lo, int skip_hi, float* out,
                                             void* stream) {
    SSAD_CHECK_ARG(img && wf && out, "null pointer");
    SSAD_CHECK_ARG(skip_lo > skip_hi || (skip_lo >= 0 && skip_hi < 16), "skipped square outside the 16 x 16 map");
    SSAD_CHECK_ARG(B > 0 && H >= 32 && W >= 32 && patch_stride > 0, "bad shape (32x32 windows)");
    StemPatchParams p;
    p.img = img; p.wf = wf; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.ps = patch_stride; p.hwnc = hwnc; p.skip_lo = skip_lo; p.skip_hi = skip_hi;
    p.prow = (H - 32) / patch_stride + 1;
    p.pcol = (W - 32) / patch_stride + 1;
    p.Nsamp = (int64_t)B * p.prow * p.pcol;
    constexpr int lds_bytes = (SP_WF + SP_SRC + SP_CB) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        SSAD_SET_DYN_LDS(stem_patch_fused_kernel, lds_bytes);
        attr_set = true;
    }
    const int64_t grid = p.Nsamp < 2048 ? p.Nsamp : 2048;
    hipLaunchKernelGGL(stem_patch_fused_kernel, dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, p);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// Pooled rows / columns 0, 1 and 15 of every patch's stem map, position-major [16][16][Nsamp][64]; the other positions are not written
// (ssad_patch_gather_hwnc_band fills rows / columns 2-3 and 13-14 from the per-image map, the interior is read by nobody).
extern "C" int ssad_stem_patch_border_fwd(const float* img, int B, int H, int W, int patch_stride, const float* wf, const float* scale,
                                          const float* shift, float* out, void* stream) {
    SSAD_CHECK_ARG(img && wf && out, "null pointer");
    SSAD_CHECK_ARG(B > 0 && H >= 32 && W >= 32 && patch_stride > 0, "bad shape (32x32 windows)");
    StemPatchParams p;
    p.img = img; p.wf = wf; p.scale = scale; p.shift = shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.ps = patch_stride; p.hwnc = 1; p.skip_lo = 1; p.skip_hi = 0;
    p.prow = (H - 32) / patch_stride + 1;
    p.pcol = (W - 32) / patch_stride + 1;
    p.Nsamp = (int64_t)B * p.prow * p.pcol;
    constexpr int lds_bytes = (SP_WF + SP_SRC + SB_CB) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        SSAD_SET_DYN_LDS(stem_patch_border_kernel, lds_bytes);
        attr_set = true;
    }
    const int64_t grid = p.Nsamp < 2048 ? p.Nsamp : 2048;
    hipLaunchKernelGGL(stem_patch_border_kernel, dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, p);
    SSAD_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int maxpool_fwd_impl(const T* in, T* out, uint8_t* idx, int64_t N, int H, int W, int C, int hwnc, void* stream,
                            PoolBN bn = PoolBN{nullptr, nullptr, nullptr, nullptr}, T* zwin = nullptr) {
    SSAD_CHECK_ARG(in && out, "null pointer");
    SSAD_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bad shape (C % 4)");
    int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    constexpr int E = Lane<T>::E;
    SSAD_CHECK_ARG(C % E == 0, "channel count must be a multiple of the 16-byte lane width");
    int64_t totalE = N * Ho * Wo * (C / E);
    SSAD_CHECK_ARG(cdiv64(totalE, 256) < (int64_t)2147483647, "too large");
    hipLaunchKernelGGL(maxpool3x3s2_kernel<T>, dim3((unsigned)cdiv64(totalE, 256)), dim3(256), 0, (hipStream_t)stream, in, out, idx,
                       totalE, H, W, C / E, Ho, Wo, N, hwnc, bn, zwin);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// Stem tail of the training forward in one pass: BatchNorm (batch statistics already known) + ReLU + max-pool 3x3/2
// over the raw conv1 output z; writes the pooled map and the argmax slots only.
extern "C" int ssad_bn_relu_maxpool_fwd(const float* z, const float* mean, const float* invstd, const float* gamma,
                                        const float* beta, float* out, uint8_t* idx, int64_t N, int H, int W, int C,
                                        void* stream) {
    SSAD_CHECK_ARG(mean && invstd && gamma && beta && idx, "null pointer");
    return maxpool_fwd_impl<float>(z, out, idx, N, H, W, C, 0, stream, PoolBN{mean, invstd, gamma, beta});
}

// the same over tensors stored as halves (precision-16 step: z, the pooled map)
extern "C" int ssad_bn_relu_maxpool_fwd_h(const void* z, const float* mean, const float* invstd, const float* gamma,
                                          const float* beta, void* out, uint8_t* idx, int64_t N, int H, int W, int C,
                                          void* stream) {
    SSAD_CHECK_ARG(mean && invstd && gamma && beta && idx, "null pointer");
    return maxpool_fwd_impl<hf>((const hf*)z, (hf*)out, idx, N, H, W, C, 0, stream, PoolBN{mean, invstd, gamma, beta});
}

// ... also writing zwin[N][Ho][Wo][C]: the RAW z of each window's winner.  The BatchNorm backward reduction of the stem then runs over the
// POOLED tensors (sum over pixels of g and g * xhat = sum over windows of dpool * mask(zwin) and dpool * mask * xhat(zwin): every window
// routes its gradient to exactly one pixel) -- 1/4 of the rows and no pass over the 128 x 128 z (ssad_bn_bwd_reduce_zmask on (dpool, zwin),
// then ssad_pool_bn_relu_bwd_apply).
extern "C" int ssad_bn_relu_maxpool_fwd_win(const float* z, const float* mean, const float* invstd, const float* gamma,
                                            const float* beta, float* out, uint8_t* idx, float* zwin, int64_t N, int H, int W, int C,
                                            void* stream) {
    SSAD_CHECK_ARG(mean && invstd && gamma && beta && idx && zwin, "null pointer");
    return maxpool_fwd_impl<float>(z, out, idx, N, H, W, C, 0, stream, PoolBN{mean, invstd, gamma, beta}, zwin);
}
extern "C" int ssad_bn_relu_maxpool_fwd_win_h(const void* z, const float* mean, const float* invstd, const float* gamma,
                                              const float* beta, void* out, uint8_t* idx, void* zwin, int64_t N, int H, int W, int C,
                                              void* stream) {
    SSAD_CHECK_ARG(mean && invstd && gamma && beta && idx && zwin, "null pointer");
    return maxpool_fwd_impl<hf>((const hf*)z, (hf*)out, idx, N, H, W, C, 0, stream, PoolBN{mean, invstd, gamma, beta}, (hf*)zwin);
}

extern "C" int ssad_maxpool3x3s2_fwd(const float* in, float* out, int64_t N, int H, int W, int C, int hwnc, void* stream) {
    return maxpool_fwd_impl<float>(in, out, nullptr, N, H, W, C, hwnc, stream);
}

// Training forward: also records, per output element, which of the 9 window slots held the first maximum
// (one byte each, [N][Ho][Wo][C]); ssad_maxpool3x3s2_bwd_idx routes gradients with it.
extern "C" int ssad_maxpool3x3s2_fwd_idx(const float* in, float* out, uint8_t* idx, int64_t N, int H, int W, int C, void* stream) {
    SSAD_CHECK_ARG(idx, "null index buffer");
    return maxpool_fwd_impl<float>(in, out, idx, N, H, W, C, 0, stream);
}
