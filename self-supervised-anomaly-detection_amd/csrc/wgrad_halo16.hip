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
// (ky) is a whole halo row and keeps 16-byte alignment, the column part (kx) does not -- so a lane reads, per (K-step, ky), the aligned
// 16 bytes of columns 0-7 of its row piece plus the 4 bytes of columns 8-9, and forms the three kx fragments in registers: kx = 0 as
// read, kx = 2 by renaming dwords, kx = 1 with four v_alignbit_b32 (the matrix waves have the issue slots to spare):
//   A (dZ^T): dzT[co][py * TW + px]                                   72-half rows (144 B: conflict-free b128 reads over 16 lanes)
//   B (X^T) : xT[ci][hy * RP + hx] = X[y0 - 1 + hy][x0 - 1 + hx]      rows of RP = 24 (TW = 16) or 16 (TW = 8) halves, 152 / 168 halves
//             per channel (304 / 336 B: likewise conflict-free)
// 29 KB (31 KB) of LDS per stage, two stages, one workgroup of eight waves per CU (four matrix waves, four staging waves); nine
// 32 x 32 accumulators per matrix wave (144 registers) across all tiles of the workgroup; per-workgroup blocks go to slab[split] and
// ssad_wgrad_reduce sums the splits in a fixed order, exactly as for the fp32 kernel.  Zero padding and ragged tiles are staged as zeros.
//
// Replaces the same autograd node as wgrad.hip's 16-bit kernels (conv2d weight gradient under loss.backward() with autocast,
// tools.py:263, :270, :303); the stride-2 layers and the 1 x 1 layers stay on wgrad_bf16_kernel.
#include "common.h"
#include <stdlib.h>

namespace {

struct WgH16Params {
    const void* dz;      // [N][H][W][Cout]  floats, or halves (TI = hf: the precision-16 step with half tensors)
    const void* x;       // [N][H][W][Cin]
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

// 512 threads: waves 0-3 run the matrix instructions, waves 4-7 only stage (fetch -> convert -> transposed LDS stores), one tile
// ahead in LDS (two stages) and one more in registers.  With MFMAs this short a tile is 0.5 us of matrix work against 2-4 us of
// memory latency: a workgroup that does both in turn spends its life waiting (first version of this kernel: 8 us per tile per
// workgroup, 210-265 us per layer; this form 115-140 us).  What bounds it now is instruction issue: every SIMD carries one matrix wave
// (0.83 us per tile, 0.48 of it MFMAs) and one staging wave (0.5-1 us), and a wave that streams MFMAs keeps the SIMD's issue to
// itself, so the two add up (measured by switching either side off: 82 us matrix waves alone, 93-129 us staging waves alone, of which
// 29 us are the slab epilogue and its reduction).  Tried on top, no gain: a second register set so that loads are issued two
// barriers ahead, XCD-contiguous numbering of the blocks that walk the same tiles, fragment reads one K-step ahead (slower).
template <int TH, int TW, bool F16, typename TI = float>
__global__ __launch_bounds__(512, 1) void wgrad3x3_halo16_kernel(WgH16Params p) {
    using op_t = typename Op16<F16>::t;
    using op4 = typename Op16<F16>::v4;
    using op8 = typename Op16<F16>::v8;
    constexpr int P = TH * TW, HH = TH + 2;
    static_assert(P == 64 && TW % 8 == 0, "64-pixel tiles, rows of 8 or 16 pixels");
    constexpr int DP = P + 8;                   // halves per dzT row
    constexpr int RP = TW + 8;                  // halves per halo row: TW + 2 pixels, padded to a multiple of 8
    constexpr int XP = HH * RP + 8;             // halves per channel of the halo
    constexpr int GR = (TW + 2 + 3) / 4;        // 4-pixel groups per halo row (the last one partly beyond the halo: stored as read)
    constexpr int NXB = HH * GR * 16;           // (halo row, pixel group, channel quad) blocks
    constexpr int NXI = (NXB + 255) / 256;
    constexpr int STAGE = 64 * DP + 64 * XP;    // halves per stage: dzT [64 co][DP], then xT [64 ci][XP]
    extern __shared__ __attribute__((aligned(16))) float lds[];
    op_t* L = (op_t*)lds;
    const int tid = threadIdx.x;
    const bool loader = tid >= 256;
    const int lt = tid & 255, lane = tid & 63, wave = (tid >> 6) & 3;
    const int r = lane & 31, h = lane >> 5;
    const int cb = wave & 1, ib = wave >> 1;    // 32-wide co / ci block of this wave inside the 64 x 64 block

    const int pair = blockIdx.x % p.npairs, split = blockIdx.x / p.npairs;
    const int co0 = (pair / p.ci_tiles) * 64, ci0 = (pair % p.ci_tiles) * 64;
    const int64_t t_begin = (int64_t)split * p.chunk;
    const int64_t t_end = t_begin + p.chunk < p.ntiles ? t_begin + p.chunk : p.ntiles;
    const int nt = (int)(t_end - t_begin);
    const int tpi = p.tiles_y * p.tiles_x;

    if (loader) {
        const int pg = lt & 15, cq = lt >> 4;   // dZ: 4-pixel group, channel quad of this thread
        using raw4 = typename Raw4<TI>::t;          // fp32 tensors: four floats; half tensors: the four halves as stored
        const raw4 zero4 = {0, 0, 0, 0};
        raw4 dv[4], xv[NXI][4];
        // Everything about a thread's pieces that does not depend on the tile is computed once: position inside the tile / halo and the
        // offset from the tile's first pixel.  Tiles are taken in order, so (n, ty, tx) advance by counting -- the staging waves' own
        // instruction count is what bounds this kernel once the loads are far enough ahead.
        const int dzy = (pg * 4) / TW, dzx = (pg * 4) % TW;
        const unsigned dz_off = (unsigned)((dzy * p.W + dzx) * p.Cout + cq * 4);
        int xhy[NXI], xhx[NXI];
        unsigned x_off[NXI];
        bool x_on[NXI];
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int b = lt + 256 * i;
            const int xq = b & 15, g = (b >> 4) % GR;
            xhy[i] = (b >> 4) / GR;
            xhx[i] = g * 4;
            x_on[i] = b < NXB;
            x_off[i] = (unsigned)((xhy[i] * p.W + xhx[i]) * p.Cin + xq * 4);
        }
        int ld_n = (int)(t_begin / tpi), ld_ty, ld_tx;
        {
            const int rem = (int)(t_begin - (int64_t)ld_n * tpi);
            ld_ty = rem / p.tiles_x;
            ld_tx = rem - ld_ty * p.tiles_x;
        }
        auto load_tile = [&]() {                 // the next tile of this workgroup: all its loads in flight together
            const int y0 = ld_ty * TH, x0 = ld_tx * TW;
            const TI* dzn = (const TI*)p.dz + ((int64_t)ld_n * p.H * p.W + (int64_t)y0 * p.W + x0) * p.Cout + co0;
            // first halo pixel (y0 - 1, x0 - 1): outside the image on border tiles -- never dereferenced there
            const TI* xn = (const TI*)p.x + ((int64_t)ld_n * p.H * p.W + (int64_t)(y0 - 1) * p.W + (x0 - 1)) * p.Cin + ci0;
            const bool dy_ok = y0 + dzy < p.H;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                raw4 v = zero4;
                if (dy_ok && x0 + dzx + q < p.W) v = ldraw4(dzn + dz_off + (unsigned)(q * p.Cout));
                dv[q] = v;
            }
#pragma unroll
            for (int i = 0; i < NXI; ++i) {
                const bool row_ok = x_on[i] && (unsigned)(y0 - 1 + xhy[i]) < (unsigned)p.H;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    raw4 v = zero4;
                    if (row_ok && xhx[i] + q < TW + 2 && (unsigned)(x0 - 1 + xhx[i] + q) < (unsigned)p.W)
                        v = ldraw4(xn + x_off[i] + (unsigned)(q * p.Cin));
                    xv[i][q] = v;
                }
            }
            if (++ld_tx == p.tiles_x) { ld_tx = 0; if (++ld_ty == p.tiles_y) { ld_ty = 0; ++ld_n; } }
        };
        auto store_tile = [&](op_t* st) {        // convert, transpose: four 8-byte column pieces per 4 x 4 block
            op_t* dzT = st;
            op_t* xT = st + 64 * DP;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const op4 v = {(op_t)dv[0][k], (op_t)dv[1][k], (op_t)dv[2][k], (op_t)dv[3][k]};
                *(op4*)(dzT + (cq * 4 + k) * DP + pg * 4) = v;
            }
#pragma unroll
            for (int i = 0; i < NXI; ++i) {
                const int b = lt + 256 * i;
                if (b >= NXB) continue;
                const int xq = b & 15, g = (b >> 4) % GR, hy = (b >> 4) / GR;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const op4 v = {(op_t)xv[i][0][k], (op_t)xv[i][1][k], (op_t)xv[i][2][k], (op_t)xv[i][3][k]};
                    *(op4*)(xT + (xq * 4 + k) * XP + hy * RP + g * 4) = v;
                }
            }
        };
        if (nt > 0) {
            load_tile();
            store_tile(L);
            if (nt > 1) load_tile();
        }
        __syncthreads();
        for (int s = 0; s < nt; ++s) {
            if (s + 1 < nt) store_tile(L + ((s + 1) & 1) * STAGE);      // the stage the matrix waves left at the previous barrier
            if (s + 2 < nt) load_tile();
            __syncthreads();
        }
        return;
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __syncthreads();
    for (int s = 0; s < nt; ++s) {
        const op_t* st = L + (s & 1) * STAGE;
        const op_t* ap = st + (cb * 32 + r) * DP + 8 * h;
        // lane half h: pixels 8 h .. 8 h + 7 of tile row k (TW = 16), or the whole tile row 2 k + h (TW = 8)
        const op_t* bp = st + 64 * DP + (ib * 32 + r) * XP + (TW == 16 ? 8 * h : RP * h);
        // 4 K-steps of 16 pixels x 9 taps.  A = dZ^T[co][16 k + 8 h ..]; per ky one aligned 16-byte piece of X^T + the next dword,
        // from which the three kx fragments are formed in registers
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const op8 a = *(const op8*)(ap + k * 16);
            u32x4 c0[3];
            unsigned c1[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int row = TW == 16 ? k + ky : 2 * k + ky;
                c0[ky] = *(const u32x4*)(bp + row * RP);
                c1[ky] = *(const unsigned*)(bp + row * RP + 8);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const u32x4 b1 = {__builtin_amdgcn_alignbit(c0[ky][1], c0[ky][0], 16), __builtin_amdgcn_alignbit(c0[ky][2], c0[ky][1], 16),
                                  __builtin_amdgcn_alignbit(c0[ky][3], c0[ky][2], 16), __builtin_amdgcn_alignbit(c1[ky], c0[ky][3], 16)};
                const u32x4 b2 = {c0[ky][1], c0[ky][2], c0[ky][3], c1[ky]};
                acc[ky * 3 + 0] = Op16<F16>::mfma(a, __builtin_bit_cast(op8, c0[ky]), acc[ky * 3 + 0]);
                acc[ky * 3 + 1] = Op16<F16>::mfma(a, __builtin_bit_cast(op8, b1), acc[ky * 3 + 1]);
                acc[ky * 3 + 2] = Op16<F16>::mfma(a, __builtin_bit_cast(op8, b2), acc[ky * 3 + 2]);
            }
        }
        __syncthreads();
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
    // one workgroup per CU in one round of equal work; every split adds one slab (written once, read once by the reduction)
    static const int target = getenv("SSAD_WGRAD_HALO16_WGS") ? atoi(getenv("SSAD_WGRAD_HALO16_WGS")) : 256;
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
static int wgrad3x3_halo16_impl(const void* dz, const void* x, int half_in, float* slab, int splits, int64_t N, int H, int W, int Cin,
                                int Cout, int f16, int64_t dz_elems, void* stream) {
    SSAD_CHECK_ARG(dz && x && slab && N > 0 && H > 0 && W > 0, "bad argument");
    SSAD_CHECK_ARG(dz_elems == N * H * W * Cout, "dz does not hold N x H x W x Cout elements");
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
        constexpr int bytes = 2 * (64 * (64 + 8) + 64 * (6 * 24 + 8)) * 2;
        static bool set = false;
        if (!set) {
            SSAD_SET_DYN_LDS((wgrad3x3_halo16_kernel<4, 16, true>), bytes);
            SSAD_SET_DYN_LDS((wgrad3x3_halo16_kernel<4, 16, false>), bytes);
            SSAD_SET_DYN_LDS((wgrad3x3_halo16_kernel<4, 16, true, hf>), bytes);
            set = true;
        }
        if (half_in) hipLaunchKernelGGL((wgrad3x3_halo16_kernel<4, 16, true, hf>), dim3(grid), dim3(512), bytes, st, p);
        else if (f16) hipLaunchKernelGGL((wgrad3x3_halo16_kernel<4, 16, true>), dim3(grid), dim3(512), bytes, st, p);
        else hipLaunchKernelGGL((wgrad3x3_halo16_kernel<4, 16, false>), dim3(grid), dim3(512), bytes, st, p);
    } else {
        constexpr int bytes = 2 * (64 * (64 + 8) + 64 * (10 * 16 + 8)) * 2;
        static bool set = false;
        if (!set) {
            SSAD_SET_DYN_LDS((wgrad3x3_halo16_kernel<8, 8, true>), bytes);
            SSAD_SET_DYN_LDS((wgrad3x3_halo16_kernel<8, 8, false>), bytes);
            SSAD_SET_DYN_LDS((wgrad3x3_halo16_kernel<8, 8, true, hf>), bytes);
            set = true;
        }
        if (half_in) hipLaunchKernelGGL((wgrad3x3_halo16_kernel<8, 8, true, hf>), dim3(grid), dim3(512), bytes, st, p);
        else if (f16) hipLaunchKernelGGL((wgrad3x3_halo16_kernel<8, 8, true>), dim3(grid), dim3(512), bytes, st, p);
        else hipLaunchKernelGGL((wgrad3x3_halo16_kernel<8, 8, false>), dim3(grid), dim3(512), bytes, st, p);
    }
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_conv_wgrad3x3_halo16(const float* dz, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin,
                                         int Cout, int f16, int64_t dz_elems, void* stream) {
    return wgrad3x3_halo16_impl(dz, x, 0, slab, splits, N, H, W, Cin, Cout, f16, dz_elems, stream);
}

// dz and x stored as halves (fp16 operands without a conversion); slabs and their reduction as above
extern "C" int ssad_conv_wgrad3x3_halo16_h(const void* dz, const void* x, float* slab, int splits, int64_t N, int H, int W, int Cin,
                                           int Cout, int64_t dz_elems, void* stream) {
    return wgrad3x3_halo16_impl(dz, x, 1, slab, splits, N, H, W, Cin, Cout, 1, dz_elems, stream);
}
