// Weight gradient of the 3x3 / pad 1 convolutions (stride 1 and 2) over HALF tensors, fp16 operands, fp32 accumulation -- the
// conv2d weight-gradient nodes of the trunk under the reference's pl.Trainer(precision=16) (tools.py:263, :270, :303):
//
//   dW[co][ky][kx][ci] = sum over output pixels p of  dZ[p][co] * X[S p + (ky - 1, kx - 1)][ci]
//
// v_mfma_f32_32x32x16_f16 contracts over 16 PIXELS and wants, per lane, 8 consecutive pixels of ONE channel -- the transpose of how the
// tensors lie in memory ([pixel][channel]).  wgrad_halo16.hip transposes while it stages (four staging waves convert / shuffle /
// store [channel][pixel] tiles; it is bound by THEIR instruction issue: 110-120 us per layer whatever the layer, against 31 us of
// matrix work).  With half tensors nothing needs converting, so here the tiles go to LDS exactly as they are fetched -- 16-byte
// pieces of [pixel][64 channels] rows, no staging waves -- and the transpose happens in the FRAGMENT READS: a lane gathers its 8
// pixels with eight 2-byte LDS reads at constant offsets (one pixel apart), which land directly in the halves of four registers
// (ds_read_u16_d16 / _d16_hi).  Per 16-pixel K-step a wave reads 8 values of dZ and, per filter row, the 10 (stride 1) or 17
// (stride 2) values of X its three kx fragments are cut from: 38 / 59 two-byte reads for 9 MFMAs.  Stride 2 -- the first conv of
// layer2-4, which had no halo form and took 140-330 us on the one-tap-per-workgroup kernel -- is the same code with pixel stride 2.
// A workgroup (four waves, 2 x 2 over a 64 x 64 (co, ci) block, nine 32 x 32 accumulators each) walks pixel tiles; the next tile is
// in flight in registers while the current one is contracted; per-workgroup blocks go to slab[split] and ssad_wgrad_reduce sums the
// splits in a fixed order, as for the other weight-gradient kernels.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int LD = 64 + 8;          // halves per LDS pixel row (144 B)

struct Wg16Params {
    const hf* dz;       // [N][Ho][Wo][Cout]
    const hf* x;        // [N][H][W][Cin]
    float* slab;        // [splits][Cout][9 * Cin]
    int N, Ho, Wo, H, W, Cin, Cout;
    int tiles_y, tiles_x, ci_tiles, npairs, splits;
    int64_t ntiles, chunk;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// S: stride.  TH x TW: output pixels per tile (TW = 16 or 8; TH * TW = 64 for stride 1, 32 for stride 2).
template <int S, int TH, int TW>
__global__ __launch_bounds__(256, 2) void wgrad3x3_g16_kernel(Wg16Params p) {
    constexpr int P = TH * TW;                      // output pixels per tile
    constexpr int KS = P / 16;                      // 16-pixel K-steps per tile
    constexpr int HH = S * (TH - 1) + 3, HW = S * (TW - 1) + 3;     // input halo of a tile
    constexpr int NHP = HH * HW;
    constexpr int NDZ = P / 32;                     // dZ pieces per thread (8 threads per pixel)
    constexpr int NX = (NHP + 31) / 32;             // X pieces per thread
    static_assert(P % 32 == 0 && (TW == 16 || TW == 8), "tile shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    hf* dzs = (hf*)lds;                             // [P][LD]
    hf* xs = dzs + P * LD;                          // [NHP][LD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cb = wave & 1, ib = wave >> 1;        // 32-wide co / ci block of this wave inside the 64 x 64 block
    const int piece = tid & 7, prow = tid >> 3;

    const int pair = blockIdx.x % p.npairs, split = blockIdx.x / p.npairs;
    const int co0 = (pair / p.ci_tiles) * 64, ci0 = (pair % p.ci_tiles) * 64;
    const int64_t t_begin = (int64_t)split * p.chunk;
    const int64_t t_end = t_begin + p.chunk < p.ntiles ? t_begin + p.chunk : p.ntiles;
    const int tpi = p.tiles_y * p.tiles_x;

    u32x4 dreg[NDZ], xreg[NX];
    auto load_tile = [&](int64_t t) {
        const int n = (int)(t / tpi);
        const int rem = (int)(t - (int64_t)n * tpi);
        const int y0 = (rem / p.tiles_x) * TH, x0 = (rem % p.tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < NDZ; ++i) {
            const int px = prow + 32 * i;
            const int y = y0 + px / TW, x = x0 + px % TW;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (y < p.Ho && x < p.Wo) v = *(const u32x4*)(p.dz + (((int64_t)n * p.Ho + y) * p.Wo + x) * p.Cout + co0 + piece * 8);
            dreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int hp = prow + 32 * i;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int y = S * y0 - 1 + hy, x = S * x0 - 1 + hx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (hp < NHP && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                v = *(const u32x4*)(p.x + (((int64_t)n * p.H + y) * p.W + x) * p.Cin + ci0 + piece * 8);
            xreg[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NDZ; ++i) *(u32x4*)(dzs + (prow + 32 * i) * LD + piece * 8) = dreg[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int hp = prow + 32 * i;
            if (hp < NHP) *(u32x4*)(xs + hp * LD + piece * 8) = xreg[i];
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    if (t_begin < t_end) load_tile(t_begin);
    for (int64_t t = t_begin; t < t_end; ++t) {
        __syncthreads();                            // the previous tile's readers are done
        store_tile();
        __syncthreads();
        if (t + 1 < t_end) load_tile(t + 1);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            // this lane's 8 pixels of the K-step: TW = 16: tile row k, columns 8 h .. 8 h + 7; TW = 8: tile row 2 k + h, columns 0 .. 7
            const int py = TW == 16 ? k : 2 * k + h, px0 = TW == 16 ? 8 * h : 0;
            const hf* ap = dzs + (py * TW + px0) * LD + cb * 32 + r;
            f16x8 a;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = ap[j * LD];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const hf* bp = xs + ((S * py + ky) * HW + S * px0) * LD + ib * 32 + r;
                constexpr int NV = S * 7 + 3;       // input columns the three kx fragments of a filter row are cut from: 10 / 17
                hf v[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] = bp[j * LD];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    f16x8 b;
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[j] = v[S * j + kx];
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[ky * 3 + kx], 0, 0, 0);
                }
            }
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

struct G16Geo {
    int TH, TW, tiles_y, tiles_x, npairs, splits;
    int64_t ntiles, chunk;
};

static G16Geo g16_geometry(int64_t N, int Ho, int Wo, int Cin, int Cout, int S) {
    G16Geo g;
    g.TW = Wo > 8 ? 16 : 8;
    g.TH = (S == 1 ? 64 : 32) / g.TW;
    g.tiles_y = (Ho + g.TH - 1) / g.TH;
    g.tiles_x = (Wo + g.TW - 1) / g.TW;
    g.ntiles = N * g.tiles_y * g.tiles_x;
    g.npairs = (Cin / 64) * (Cout / 64);
    // two workgroups per CU in one round of equal work; every split adds one slab (written once, read once by the reduction)
    static const int target = getenv("SSAD_WGRAD_G16_WGS") ? atoi(getenv("SSAD_WGRAD_G16_WGS")) : 512;
    int64_t s = (target + g.npairs - 1) / g.npairs;
    if (s > g.ntiles / 4) s = g.ntiles / 4;
    if (s < 1) s = 1;
    g.splits = (int)s;
    g.chunk = (g.ntiles + g.splits - 1) / g.splits;
    return g;
}

template <int S, int TH, int TW>
static int g16_launch(const Wg16Params& p, unsigned grid, hipStream_t st) {
    constexpr int HH = S * (TH - 1) + 3, HW = S * (TW - 1) + 3;
    constexpr int bytes = (TH * TW + HH * HW) * LD * 2;
    hipLaunchKernelGGL((wgrad3x3_g16_kernel<S, TH, TW>), dim3(grid), dim3(256), bytes, st, p);
    return 0;
}

}  // namespace

// 1 when ssad_conv_wgrad3x3_g16_h handles the layer: 3 x 3, pad 1, stride 1 or 2, channel counts multiples of 64.
extern "C" int ssad_wgrad3x3_g16_ok(int Cin, int Cout, int KH, int KW, int stride, int pad) {
    static const int on = getenv("SSAD_WGRAD_G16") ? atoi(getenv("SSAD_WGRAD_G16")) : 1;
    return on && KH == 3 && KW == 3 && pad == 1 && (stride == 1 || stride == 2) && Cin % 64 == 0 && Cout % 64 == 0;
}

// Ho, Wo: size of dz (the conv's OUTPUT)
extern "C" int ssad_wgrad3x3_g16_splits(int64_t N, int Ho, int Wo, int Cin, int Cout, int stride) {
    return g16_geometry(N, Ho, Wo, Cin, Cout, stride).splits;
}

// dz NHWC halves [N][Ho][Wo][Cout], x NHWC halves [N][H][W][Cin] (3 x 3, pad 1, stride 1 or 2: Ho = (H - 1) / stride + 1) ->
// slab[splits][Cout][9 * Cin] fp32 with splits = ssad_wgrad3x3_g16_splits(...); follow with ssad_wgrad_reduce(slab, dw, splits, Cout,
// 9 * Cin, 3, 3, Cin, ...).  dz_elems: what the caller's dz buffer holds (guard convention of include/ssad.h).
extern "C" int ssad_conv_wgrad3x3_g16_h(const void* dz, const void* x, float* slab, int splits, int64_t N, int Ho, int Wo, int H, int W,
                                        int Cin, int Cout, int stride, int64_t dz_elems, void* stream) {
    SSAD_CHECK_ARG(dz && x && slab && N > 0 && H > 0 && W > 0, "bad argument");
    SSAD_CHECK_ARG(stride == 1 || stride == 2, "stride 1 or 2");
    SSAD_CHECK_ARG(Cin % 64 == 0 && Cout % 64 == 0, "channel counts must be multiples of 64");
    SSAD_CHECK_ARG(Ho == (H - 1) / stride + 1 && Wo == (W - 1) / stride + 1, "dz / x sizes disagree for a 3 x 3 / pad 1 conv of this stride");
    SSAD_CHECK_ARG(dz_elems == N * Ho * Wo * Cout, "dz does not hold N x Ho x Wo x Cout elements");
    const G16Geo g = g16_geometry(N, Ho, Wo, Cin, Cout, stride);
    SSAD_CHECK_ARG(splits == g.splits, "splits must come from ssad_wgrad3x3_g16_splits");
    SSAD_CHECK_ARG(g.ntiles < (int64_t)1 << 31, "too many tiles");
    Wg16Params p;
    p.dz = (const hf*)dz; p.x = (const hf*)x; p.slab = slab;
    p.N = (int)N; p.Ho = Ho; p.Wo = Wo; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.tiles_y = g.tiles_y; p.tiles_x = g.tiles_x; p.ci_tiles = Cin / 64; p.npairs = g.npairs; p.splits = g.splits;
    p.ntiles = g.ntiles; p.chunk = g.chunk;
    const unsigned grid = (unsigned)(g.npairs * g.splits);
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1) {
        if (g.TW == 16) g16_launch<1, 4, 16>(p, grid, st);
        else g16_launch<1, 8, 8>(p, grid, st);
    } else {
        if (g.TW == 16) g16_launch<2, 2, 16>(p, grid, st);
        else g16_launch<2, 4, 8>(p, grid, st);
    }
    SSAD_CHECK_LAUNCH();
    return 0;
}
