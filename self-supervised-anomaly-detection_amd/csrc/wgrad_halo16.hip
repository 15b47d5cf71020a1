// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions with fp16 (or bf16) OPERANDS and fp32 accumulation -- the
// Trainer(precision=16) path, which is what the reference's pl.Trainer(precision=16) asks for (tools.py:263) -- as a HALO-TILE kernel:
//
//   dW[co][ky][kx][ci] = sum over pixels p of  f16(dZ[p][co]) * f16(X[p + (ky-1, kx-1)][ci])
//
// The split-over-pixels form (wgrad.hip, wgrad_bf16_kernel) gives a workgroup ONE filter tap: dZ and X are fetched, converted and
// transposed once per tap, and with MFMAs sixteen times shorter than the fp32 ones that staging is all the kernel does (round 4:
// 6.05 of the 20.2 ms of a precision-16 step, ~0.33 ms per layer against an HBM floor of 0.01-0.11 ms).  Here, as in wgrad_halo.hip,
// a workgroup owns a 64 x 64 (co, ci) block for ALL nine taps and walks over 64-pixel tiles (4 x 16, or 8 x 8 on maps up to 8 wide):
// the dZ tile and the halo of X are fetched and converted once per tile.
//
// v_mfma_f32_32x32x16_f16 contracts over 16 PIXELS per instruction and wants, per lane, 8 consecutive pixels of one channel: the tiles
// live TRANSPOSED in LDS, [channel][pixel] halves.  A filter tap shifts the pixels an X fragment starts at; the row part of the shift
// (ky) is a multiple of the tile width and keeps 16-byte alignment, the column part (kx) does not -- so the halo is kept in THREE copies,
// copy kx holding columns kx .. kx + TW - 1, and every fragment read is an aligned ds_read_b128 at an immediate offset:
//   A (dZ^T): dzT[co][py * TW + px]                      72-half rows  (144 B: conflict-free b128 reads over 16 lanes)
//   B (X^T) : xT[kx][ci][hy * TW + j] = X[y0 - 1 + hy][x0 - 1 + j + kx]     104 / 88-half rows (208 / 176 B: likewise)
// 49 KB (43 KB) of LDS per workgroup, two workgroups per CU; nine 32 x 32 accumulators per wave (144 registers) across all tiles of
// the workgroup; per-workgroup blocks go to slab[split] and ssad_wgrad_reduce sums the splits in a fixed order, exactly as for the
// fp32 kernel.  Zero padding and ragged tiles are staged as zeros.
//
// Replaces the same autograd node as wgrad.hip's 16-bit kernels (conv2d weight gradient under loss.backward() with autocast,
// tools.py:263, :270, :303); the stride-2 layers and the 1 x 1 layers stay on wgrad_bf16_kernel.
#include "common.h"
#include <stdlib.h>

namespace {

struct WgH16Params {
    const float* dz;     // [N][H][W][Cout]
    const float* x;      // [N][H][W][Cin]
    float* slab;         // [splits][Cout][9 * Cin]
    int N, H, W, Cin, Cout;
    int tiles_y, tiles_x, ci_tiles, npairs, splits;
    int64_t ntiles, chunk;
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

template <int TH, int TW, bool F16>
__global__ __launch_bounds__(256, 2) void wgrad3x3_halo16_kernel(WgH16Params p) {
    using op_t = typename Op16<F16>::t;
    using op4 = typename Op16<F16>::v4;
    using op8 = typename Op16<F16>::v8;
    constexpr int P = TH * TW, HH = TH + 2;
    static_assert(P == 64 && TW % 8 == 0, "64-pixel tiles, rows of 8 or 16 pixels");
    constexpr int DP = P + 8;                   // halves per dzT row
    constexpr int XP = HH * TW + 8;             // halves per xT row
    constexpr int XG = HH * TW / 4;             // 4-pixel groups of one halo copy
    constexpr int NXB = 3 * XG * 16;            // (copy, pixel group, channel quad) blocks of the halo
    constexpr int NXI = (NXB + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    op_t* dzT = (op_t*)lds;                     // [64 co][DP]
    op_t* xT = dzT + 64 * DP;                   // [3][64 ci][XP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cb = wave & 1, ib = wave >> 1;    // 32-wide co / ci block of this wave inside the 64 x 64 block

    const int pair = blockIdx.x % p.npairs, split = blockIdx.x / p.npairs;
    const int co0 = (pair / p.ci_tiles) * 64, ci0 = (pair % p.ci_tiles) * 64;
    const int64_t t_begin = (int64_t)split * p.chunk;
    const int64_t t_end = t_begin + p.chunk < p.ntiles ? t_begin + p.chunk : p.ntiles;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int tpi = p.tiles_y * p.tiles_x;
    const int pg = tid & 15, cq = tid >> 4;     // dZ: 4-pixel group, channel quad of this thread
    const op_t* ap = dzT + (cb * 32 + r) * DP + 8 * h;
    const op_t* bp = xT + (ib * 32 + r) * XP + (TW == 16 ? 8 * h : TW * h);

    for (int tile = (int)t_begin; tile < (int)t_end; ++tile) {
        const int n = tile / tpi;
        const int rem = tile - n * tpi;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        const float* dzn = p.dz + (int64_t)n * p.H * p.W * p.Cout + co0 + cq * 4;
        const float* xn = p.x + (int64_t)n * p.H * p.W * p.Cin + ci0;

        // ---- fetch (all loads of the tile in flight together), convert, transpose ----
        f32x4 dv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tp = pg * 4 + q;
            const int y = y0 + tp / TW, x = x0 + tp % TW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y < p.H && x < p.W) v = *(const f32x4*)(dzn + (unsigned)((y * p.W + x) * p.Cout));
            dv[q] = v;
        }
        f32x4 xv[NXI][4];
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int b = tid + 256 * i;
            const int xq = b & 15, g = (b >> 4) % XG, c = (b >> 4) / XG;
            const int hy = (g * 4) / TW, j0 = (g * 4) % TW;
            const int y = y0 - 1 + hy;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int x = x0 - 1 + j0 + q + c;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (b < NXB && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                    v = *(const f32x4*)(xn + (unsigned)((y * p.W + x) * p.Cin + xq * 4));
                xv[i][q] = v;
            }
        }
        __syncthreads();                        // the previous tile's fragments have all been read
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const op4 v = {(op_t)dv[0][k], (op_t)dv[1][k], (op_t)dv[2][k], (op_t)dv[3][k]};
            *(op4*)(dzT + (cq * 4 + k) * DP + pg * 4) = v;
        }
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int b = tid + 256 * i;
            if (b >= NXB) continue;
            const int xq = b & 15, g = (b >> 4) % XG, c = (b >> 4) / XG;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const op4 v = {(op_t)xv[i][0][k], (op_t)xv[i][1][k], (op_t)xv[i][2][k], (op_t)xv[i][3][k]};
                *(op4*)(xT + (c * 64 + xq * 4 + k) * XP + g * 4) = v;
            }
        }
        __syncthreads();

        // ---- 4 K-steps of 16 pixels x 9 taps.  Lane (r, h): A = dZ^T[co][16 s + 8 h ..], B_tap = X^T copy kx, row shifted by ky ----
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const op8 a = *(const op8*)(ap + s * 16);
            op8 b[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - 3 * ky;
                const int row = TW == 16 ? s + ky : 2 * s + ky;          // (TW == 8: the lane half adds its own row, folded into bp)
                b[t] = *(const op8*)(bp + kx * 64 * XP + row * TW);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = Op16<F16>::mfma(a, b[t], acc[t]);
        }
    }

    // D[row = co][col = ci]: reg e of lane (r, h) = co (e & 3) + 8 (e >> 2) + 4 h, ci r
    float* out = p.slab + (int64_t)split * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            out[((int64_t)co * 9 + t) * p.Cin + ci0 + ib * 32 + r] = acc[t][e];
        }
}

static int halo16_splits(int64_t ntiles, int npairs) {
    // two workgroups per CU in one round of equal work; every split adds one slab (written once, read once by the reduction)
    static const int target = getenv("SSAD_WGRAD_HALO16_WGS") ? atoi(getenv("SSAD_WGRAD_HALO16_WGS")) : 512;
    int64_t s = (target + npairs - 1) / npairs;
    if (s > ntiles / 4) s = ntiles / 4;
    if (s < 1) s = 1;
    return (int)s;
}

}  // namespace

// 1 when ssad_conv_wgrad3x3_halo16 handles the layer (3 x 3, stride 1, pad 1, channel counts multiples of 64).
extern "C" int ssad_wgrad3x3_halo16_ok(int Cin, int Cout, int KH, int KW, int stride, int pad) {
    static const int on = getenv("SSAD_WGRAD_HALO16") ? atoi(getenv("SSAD_WGRAD_HALO16")) : 1;
    return on && KH == 3 && KW == 3 && pad == 1 && stride == 1 && Cin % 64 == 0 && Cout % 64 == 0;
}

extern "C" int ssad_wgrad3x3_halo16_splits(int64_t N, int H, int W, int Cin, int Cout) {
    const int TW = W > 8 ? 16 : 8, TH = W > 8 ? 4 : 8;
    const int64_t ntiles = N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    return halo16_splits(ntiles, (Cin / 64) * (Cout / 64));
}

// dz NHWC [N][H][W][Cout], x NHWC [N][H][W][Cin] (3x3, stride 1, pad 1), both fp32 in memory and rounded to fp16 (f16 != 0) or
// bf16 while staged -> slab[splits][Cout][9 * Cin] fp32 with splits = ssad_wgrad3x3_halo16_splits(...); follow with
// ssad_wgrad_reduce(slab, dw, splits, Cout, 9 * Cin, 3, 3, Cin, ...).
extern "C" int ssad_conv_wgrad3x3_halo16(const float* dz, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin,
                                         int Cout, int f16, void* stream) {
    SSAD_CHECK_ARG(dz && x && slab && N > 0 && H > 0 && W > 0, "bad argument");
    SSAD_CHECK_ARG(Cin % 64 == 0 && Cout % 64 == 0, "channel counts must be multiples of 64");
    SSAD_CHECK_ARG((int64_t)H * W * Cin < (int64_t)1 << 32 && (int64_t)H * W * Cout < (int64_t)1 << 32 &&
                   N * (int64_t)((H + 3) / 4) * ((W + 7) / 8) < (int64_t)1 << 31, "offsets inside an image are 32-bit, tile numbers int");
    const int TW = W > 8 ? 16 : 8, TH = W > 8 ? 4 : 8;
    WgH16Params p;
    p.dz = dz; p.x = x; p.slab = slab; p.N = (int)N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.tiles_y = (H + TH - 1) / TH; p.tiles_x = (W + TW - 1) / TW;
    p.ci_tiles = Cin / 64; p.npairs = (Cin / 64) * (Cout / 64);
    p.ntiles = N * p.tiles_y * p.tiles_x;
    SSAD_CHECK_ARG(splits >= 1 && splits == halo16_splits(p.ntiles, p.npairs), "splits must come from ssad_wgrad3x3_halo16_splits");
    p.splits = splits;
    p.chunk = (p.ntiles + splits - 1) / splits;
    const unsigned grid = (unsigned)(p.npairs * splits);
    hipStream_t st = (hipStream_t)stream;
    if (TW == 16) {
        constexpr int bytes = (64 * (64 + 8) + 3 * 64 * (6 * 16 + 8)) * 2;
        if (f16) hipLaunchKernelGGL((wgrad3x3_halo16_kernel<4, 16, true>), dim3(grid), dim3(256), bytes, st, p);
        else hipLaunchKernelGGL((wgrad3x3_halo16_kernel<4, 16, false>), dim3(grid), dim3(256), bytes, st, p);
    } else {
        constexpr int bytes = (64 * (64 + 8) + 3 * 64 * (10 * 8 + 8)) * 2;
        if (f16) hipLaunchKernelGGL((wgrad3x3_halo16_kernel<8, 8, true>), dim3(grid), dim3(256), bytes, st, p);
        else hipLaunchKernelGGL((wgrad3x3_halo16_kernel<8, 8, false>), dim3(grid), dim3(256), bytes, st, p);
    }
    SSAD_CHECK_LAUNCH();
    return 0;
}
