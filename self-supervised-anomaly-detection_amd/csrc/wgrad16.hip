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

#ifndef WG16_ABL         // timing ablations (tools/micro/wgrad16_ablate.hip): 1 = no global loads, 2 = no MFMAs, 4 = no LDS tile writes
#define WG16_ABL 0
#endif

namespace {

constexpr int LD = 96;              // halves per LDS pixel row (192 B = 48 banks: the four rows of a transposed read and the
                                    // two 16-channel halves of a 32-lane group land on eight disjoint 8-bank runs)

// ds_read_b64_tr_b16 (gfx950): per group of 16 lanes a 4-row x 16-column block of halves is read and delivered column-major --
// lane 4 q + p supplies the address of row q, columns 4 p .. 4 p + 3; lane i receives column i, rows 0 .. 3 in its four halves.
typedef __fp16 fp16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ f16x4 tr_read(const hf* p) {
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4v*)p));
}

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
    // two stages of ([P][LD] dZ, [NHP][LD] X): tile t + 1 is written while tile t is contracted -- one barrier per tile
    constexpr int STAGE = (P + NHP) * LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cb = wave & 1, ib = wave >> 1;        // 32-wide co / ci block of this wave inside the 64 x 64 block
    const int piece = tid & 7, prow = tid >> 3;
    // transposed-read roles: group g of 16 lanes = (pixel half g >> 1, channel half g & 1); lane 4 q + p of it addresses pixel q, channels 4 p ..
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int abase = (8 * (tg >> 1) + tq) * LD + cb * 32 + 16 * (tg & 1) + 4 * tp;
    const int bbase = (TW == 16 ? S * (8 * (tg >> 1) + tq) : S * (tg >> 1) * (S * (TW - 1) + 3) + S * tq) * LD + ib * 32 + 16 * (tg & 1) + 4 * tp;

    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each XCD with its own L2.  The npairs workgroups of one split read
    // the SAME pixel tiles (each its own 64 x 64 channel block) at the same pace: all of them are placed on one XCD (split % 8), so the
    // tiles come from HBM once per split and from that XCD's L2 for the other pairs (the launches move ~5.4 x the tensors otherwise:
    // 63 us of the 73 without a single MFMA, tools/micro/wgrad16_ablate.hip)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int pair = jx % p.npairs, split = (jx / p.npairs) * 8 + xcd;
    if (split >= p.splits) return;
    const int co0 = (pair / p.ci_tiles) * 64, ci0 = (pair % p.ci_tiles) * 64;
    const int64_t t_begin = (int64_t)split * p.chunk;
    const int64_t t_end = t_begin + p.chunk < p.ntiles ? t_begin + p.chunk : p.ntiles;
    const int tpi = p.tiles_y * p.tiles_x;

    // (two register sets -- a tile's loads requested two tiles before its LDS write -- were measured: 256 VGPRs, 9.32 -> 9.47 ms per
    // step on one box; one set it stays)
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
            if (y < p.Ho && x < p.Wo && !(WG16_ABL & 1)) v = *(const u32x4*)(p.dz + (((int64_t)n * p.Ho + y) * p.Wo + x) * p.Cout + co0 + piece * 8);
            dreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int hp = prow + 32 * i;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int y = S * y0 - 1 + hy, x = S * x0 - 1 + hx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (hp < NHP && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && !(WG16_ABL & 1))
                v = *(const u32x4*)(p.x + (((int64_t)n * p.H + y) * p.W + x) * p.Cin + ci0 + piece * 8);
            xreg[i] = v;
        }
    };
    auto store_tile = [&](int stage) {
        if (WG16_ABL & 4) return;
        hf* dzs = (hf*)lds + stage * STAGE;
        hf* xs = dzs + P * LD;
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

    if (t_begin < t_end) {
        load_tile(t_begin);
        store_tile(0);
        if (t_begin + 1 < t_end) load_tile(t_begin + 1);
    }
    int stage = 0;
    for (int64_t t = t_begin; t < t_end; ++t, stage ^= 1) {
        __syncthreads();                            // stage `stage` is written; the other stage's readers (tile t - 1) are done
        if (t + 1 < t_end) {
            store_tile(stage ^ 1);                  // tile t + 1 (in registers since the previous iteration) -> the free stage
            if (t + 2 < t_end) load_tile(t + 2);
        }
        const hf* dzs = (const hf*)lds + stage * STAGE;
        const hf* xs = dzs + P * LD;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            // K-step k = tile pixels 16 k .. 16 k + 15 (TW = 16: tile row k; TW = 8: rows 2 k, 2 k + 1); lane (r, h) needs pixels
            // 16 k + 8 h .. + 7 of channel r: two transposed reads of 4 pixels each (u = 0, 1)
            f16x8 a;
            {
                const f16x4 lo = tr_read(dzs + abase + (k * 16) * LD), hi = tr_read(dzs + abase + (k * 16 + 4) * LD);
                a = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    // the input pixel under tap (ky, kx) of output pixel (py, px) is halo pixel (S py + ky, S px + kx)
                    const int o0 = TW == 16 ? ((S * k + ky) * HW + kx) * LD : ((S * 2 * k + ky) * HW + kx) * LD;
                    const f16x4 lo = tr_read(xs + bbase + o0), hi = tr_read(xs + bbase + o0 + S * 4 * LD);
                    const f16x8 b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (!(WG16_ABL & 2)) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[ky * 3 + kx], 0, 0, 0);
                    else acc[ky * 3 + kx][0] += (float)b[0] + (float)a[0];
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
    constexpr int bytes = 2 * (TH * TW + HH * HW) * LD * 2;        // two stages
    static bool attr_set = false;
    if (!attr_set) {
        SSAD_SET_DYN_LDS((wgrad3x3_g16_kernel<S, TH, TW>), bytes);
        attr_set = true;
    }
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
    const unsigned grid = (unsigned)(g.npairs * ((g.splits + 7) / 8) * 8);         // 8 XCDs x pairs x splits of that XCD (see the kernel)
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
