// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions (13 of ResNet-18's 20 trunk convolutions, 3/4 of the
// weight-gradient FLOPs) on the fp32 matrix cores, as a HALO-TILE kernel:
//
//   dW[co][ky][kx][ci] = sum over pixels p of  dZ[p][co] * X[p + (ky-1, kx-1)][ci]
//
// The split-over-pixels kernel (wgrad.hip) gives every workgroup ONE filter tap: dZ and X are re-read once per tap
// and a 64-wide tile gets 16 MFMAs per wave between barriers.  Here a workgroup owns a 64 x 64 (co, ci) block for ALL
// nine taps and walks over 64-pixel tiles (4 x 16 or 8 x 8): it stages the dZ tile and the halo of X once per tile
// (zero padding written as zeros) and every pixel pair then feeds nine MFMAs -- one dZ fragment against nine shifted X
// fragments, the shift being an LDS address immediate.  288 MFMAs per wave between barriers, the next tile's operands in
// flight in registers meanwhile, operands read from HBM / L2 ~1.6x instead of 9x.  Each wave keeps nine 32 x 32 accumulators (144 registers) across all its tiles;
// the per-workgroup blocks go to slab[split] and ssad_wgrad_reduce sums the splits in a fixed order (deterministic).
//
// Replaces the same autograd node as wgrad.hip (conv2d weight gradient under loss.backward(), tools.py:270, :303).
#include "common.h"
#include <stdlib.h>

// Ablation switches for tools/micro/wgh_ablate.hip; always 0 in the library build.
#ifndef WGH_ABL
#define WGH_ABL 0
#endif

#if WGH_ABL & 16
__device__ unsigned long long* g_wgh_trace;      // [workgroups][72]: time stamps (tools/micro/wgh_ablate.hip)
#define WGH_STAMP(i) do { if (threadIdx.x == 0 && g_wgh_trace && (i) < 64) g_wgh_trace[(size_t)blockIdx.x * 72 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define WGH_STAMP(i) do { } while (0)
#endif

namespace {

struct WgHaloParams {
    const float* dz;     // [N][H][W][Cout]
    const float* x;      // [N][Hx][Wx][Cin]   (Hx, Wx) = (H, W) for stride 1, the stride-2 conv's input size otherwise
    float* slab;         // [splits][Cout][9 * Cin]
    int N, H, W, Cin, Cout, Hx, Wx;
    int tiles_y, tiles_x, ci_tiles, npairs, splits;
    int64_t ntiles, chunk;
};

// S = 2 (round 3): the three stride-2 3x3 convolutions.  The tile is still 64 OUTPUT pixels; its X halo is the (2 TH + 1) x
// (2 TW + 1) input window those pixels touch (297 / 289 positions instead of 108 / 100: 92 KB of LDS, one workgroup per CU),
// and pixel (py, px)'s tap (ky, kx) sits at halo position (2 py + ky, 2 px + kx).  X is read ~1.2x and dZ once, where the
// split-over-pixels kernel read them once per tap.
template <int TH, int TW, int S = 1>
__global__ __launch_bounds__(256, (S == 1 || TH * TW <= 32) ? 2 : 1) void wgrad3x3_halo_kernel(WgHaloParams p) {
    constexpr int P = TH * TW, HWp = S * (TW - 1) + 3, NH = (S * (TH - 1) + 3) * HWp;
    constexpr int NDZ = P / 16;                 // 16-byte pieces of the dZ tile per thread
    constexpr int NX = (NH + 15) / 16;          // ... of the X halo
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dzt = lds;                           // [P][64]
    float* halo = lds + P * 64;                 // [NH][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cb = wave & 1, ib = wave >> 1;    // 32-wide co / ci block of this wave inside the 64 x 64 block

    const int pair = blockIdx.x % p.npairs, split = blockIdx.x / p.npairs;
    const int co0 = (pair / p.ci_tiles) * 64, ci0 = (pair % p.ci_tiles) * 64;
    const int64_t t_begin = (int64_t)split * p.chunk;
    const int64_t t_end = t_begin + p.chunk < p.ntiles ? t_begin + p.chunk : p.ntiles;
    const int c4 = tid & 15, p0 = tid >> 4;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int tpi = p.tiles_y * p.tiles_x;
    f32x4 dv[NDZ], xv[NX];
    // The operands of a tile -> registers, one 16-byte piece at a time (zero padding / ragged tiles read as zeros).
    // A wave that streams fp32 MFMAs keeps its SIMD to itself (tools/micro/partner_starve.hip: the co-resident wave issues
    // NOTHING meanwhile), so address arithmetic placed before or after the matrix loop is paid in full; placed BETWEEN the
    // wave's own MFMAs it is free (the pipe is busy 64 cycles per instruction).  The pieces of the next tile are therefore
    // issued one per MFMA group inside the loop below.
    // (round 6) The pieces are BUFFER loads: base = the tile's image (scalar registers), size = that image, offset = the tile's origin
    // (uniform, possibly negative) + this thread's constant offset inside the tile / halo.  Rows above or below the image then fall
    // outside the buffer by themselves and read zeros; only the columns left and right of the image need a test (one unsigned compare
    // against the tile's valid column range).  3-4 VALU instructions per piece where the pointer form took ~11 (bounds tests, zero
    // initialisation of the predicated-off lanes, 64-bit multiply-add, branch) -- each of them ~15-25 cycles of this wave's matrix pipe.
    constexpr unsigned OOB = 0x80000000u;
    constexpr int SRD3 = 0x00020000;
    int y0n = 0, x0n = 0;                       // origin of the tile being fetched (workgroup-uniform)
    const float *dzn = p.dz, *xn = p.x;         // its image (workgroup-uniform base pointers)
    unsigned dorg = 0, xorg = 0;                // byte offsets of the tile's first pixel / the halo's pixel (-1, -1) in their images
    unsigned dcols = 0, xlo = 0, xspan = 0;     // valid tile columns: tx < dcols;  valid halo columns: hx - xlo <= xspan
    unsigned doff[NDZ], dtx[NDZ], xoff[NX], xhx[NX];
#pragma unroll
    for (int g = 0; g < NDZ; ++g) {
        const int tp = g * 16 + p0;
        dtx[g] = (unsigned)(tp % TW);
        doff[g] = (unsigned)((((tp / TW) * p.W + tp % TW) * p.Cout + c4 * 4) * 4);
    }
#pragma unroll
    for (int q = 0; q < NX; ++q) {
        const int hp = q * 16 + p0;
        const int hy = hp / HWp, hx = hp - hy * HWp;
        xhx[q] = hp < NH ? (unsigned)hx : 0x7fffffffu;          // pieces beyond the halo: never inside the column range
        xoff[q] = (unsigned)(((hy * p.Wx + hx) * p.Cin + c4 * 4) * 4);
    }
    const int dz_bytes = (p.H * p.W * p.Cout - co0) * 4, x_bytes = (p.Hx * p.Wx * p.Cin - ci0) * 4;      // from the base to the image's end
    auto set_tile = [&](int tile) {
        const int n = tile / tpi;
        const int rem = tile - n * tpi;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        y0n = ty * TH;
        x0n = tx * TW;
        dzn = p.dz + (int64_t)n * p.H * p.W * p.Cout + co0;
        xn = p.x + (int64_t)n * p.Hx * p.Wx * p.Cin + ci0;
        dorg = (unsigned)((y0n * p.W + x0n) * p.Cout * 4);
        xorg = (unsigned)(((S * y0n - 1) * p.Wx + S * x0n - 1) * p.Cin * 4);
        dcols = (unsigned)(p.W - x0n);
        xlo = S * x0n == 0 ? 1u : 0u;                            // halo column 0 is x = -1 at the image's left edge
        xspan = (unsigned)(p.Wx - S * x0n) - xlo;                // last valid halo column: x = Wx - 1, i.e. hx = Wx - S x0n
    };
    auto load_piece = [&](int g) {              // g is a compile-time constant wherever this is called
        if (g < NDZ) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dzn, 0, dz_bytes, SRD3);
            const unsigned vo = dtx[g] < dcols ? doff[g] + dorg : OOB;
            dv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
        } else if (g < NDZ + NX) {
            const int q = g - NDZ;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, x_bytes, SRD3);
            const unsigned vo = xhx[q] - xlo <= xspan ? xoff[q] + xorg : OOB;
            xv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
        }
    };
    const int t_begin_i = (int)t_begin, t_end_i = (int)t_end;
    if (t_begin_i < t_end_i) {
        set_tile(t_begin_i);
#pragma unroll
        for (int g = 0; g < NDZ + NX; ++g) load_piece(g);
    }
    WGH_STAMP(0);
#if WGH_ABL & 16
    if (threadIdx.x == 0 && g_wgh_trace) { g_wgh_trace[(size_t)blockIdx.x * 72 + 64] = __builtin_amdgcn_s_memrealtime(); g_wgh_trace[(size_t)blockIdx.x * 72 + 66] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4); }
#endif
    for (int tile = t_begin_i; tile < t_end_i; ++tile) {
        WGH_STAMP(1 + 2 * (tile - t_begin_i));
        if (!(WGH_ABL & 2) || tile == t_begin_i) {
        __syncthreads();                        // the previous tile's fragments have all been read
#pragma unroll
        for (int q = 0; q < NDZ; ++q) *(f32x4*)(dzt + (q * 16 + p0) * 64 + c4 * 4) = dv[q];
#pragma unroll
        for (int q = 0; q < NX; ++q)
            if (q * 16 + p0 < NH) *(f32x4*)(halo + (q * 16 + p0) * 64 + c4 * 4) = xv[q];
        __syncthreads();
        }
        WGH_STAMP(2 + 2 * (tile - t_begin_i));
        // next tile (the last one re-fetches itself: no branch in the matrix loop, the data is simply not used)
        set_tile(tile + 1 < t_end_i ? tile + 1 : tile);

        // lane (r, h): k slot h = pixel (py, 2 pp + h); A = dZ[pixel][co0 + 32 cb + r], B_tap = X[pixel + tap][ci0 + 32 ib + r]
        const float* ap = dzt + h * 64 + cb * 32 + r;
        const float* bp = halo + h * S * 64 + ib * 32 + r;
        // fragments of group g + 1 are requested before the MFMAs of group g (LDS latency in the shadow as well)
        constexpr int NG = TH * (TW / 2);
        float a[2], b[2][9];
        auto read_group = [&](int g, int slot) {
            const int py = g / (TW / 2), pp = g % (TW / 2);
            a[slot] = ap[(py * TW + 2 * pp) * 64];
#pragma unroll
            for (int t = 0; t < 9; ++t) b[slot][t] = bp[((S * py + t / 3) * HWp + S * 2 * pp + t % 3) * 64];
        };
        read_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG && !(WGH_ABL & 4)) read_group(g + 1, (g + 1) & 1);      // ablation 4: every group multiplies group 0's fragments
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = mfma32(a[(WGH_ABL & 4) ? 0 : (g & 1)], b[(WGH_ABL & 4) ? 0 : (g & 1)][t], acc[t]);
            if (!(WGH_ABL & 1)) load_piece(g);       // one piece of the next tile in this group's shadow
            static_assert(NDZ + NX <= NG, "more staged pieces than MFMA groups to hide them in");
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    WGH_STAMP(62);
    // D[row = co][col = ci]: reg e of lane (r, h) = co (e & 3) + 8 (e >> 2) + 4 h, ci r
    float* out = p.slab + (int64_t)split * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            out[((int64_t)co * 9 + t) * p.Cin + ci0 + ib * 32 + r] = acc[t][e];
        }
    WGH_STAMP(63);
#if WGH_ABL & 16
    if (threadIdx.x == 0 && g_wgh_trace) g_wgh_trace[(size_t)blockIdx.x * 72 + 65] = __builtin_amdgcn_s_memrealtime();
#endif
}

static int halo_splits(int64_t ntiles, int npairs) {
    // 512 workgroups = two per CU in ONE round of equal work (no tail), at least 4 tiles per workgroup; every split adds one
    // slab (written once, read once by the reduction: 75 MB per layer at 512 workgroups)
    // (a workgroup with fewer than 16 tiles spends as long on its slab as on its MFMAs: small batches take one workgroup per
    // CU instead -- a single wave per SIMD still keeps the fp32 matrix pipe ~85 % busy)
    static const int target = getenv("SSAD_WGRAD_HALO_WGS") ? atoi(getenv("SSAD_WGRAD_HALO_WGS")) : 512;
    int64_t s = (target + npairs - 1) / npairs;
    if (s * 16 > ntiles) s = (256 + npairs - 1) / npairs;
    if (s > ntiles / 4) s = ntiles / 4;
    if (s < 1) s = 1;
    return (int)s;
}

}  // namespace

// 1 when ssad_conv_wgrad3x3_halo handles the layer.
extern "C" int ssad_wgrad3x3_halo_ok(int Cin, int Cout, int KH, int KW, int stride, int pad) {
    static const int s2 = getenv("SSAD_WGRAD_HALO_S2") ? atoi(getenv("SSAD_WGRAD_HALO_S2")) : 1;
    if (!(KH == 3 && KW == 3 && pad == 1 && Cin % 64 == 0 && Cout % 64 == 0)) return 0;
    return stride == 1 ? 1 : (stride == 2 && s2) ? 2 : 0;
}

extern "C" int ssad_wgrad3x3_halo_splits(int64_t N, int H, int W, int Cin, int Cout) {
    const int TW = W > 8 ? 16 : 8, TH = W > 8 ? 4 : 8;
    const int64_t ntiles = N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    return halo_splits(ntiles, (Cin / 64) * (Cout / 64));
}

static int halo_launch(const float* dz, const float* x, float* slab, int splits, int64_t N, int H, int W, int Hx, int Wx, int Cin,
                       int Cout, int S, int64_t dz_elems, void* stream) {
    SSAD_CHECK_ARG(dz && x && slab && N > 0 && H > 0 && W > 0, "bad argument");
    SSAD_CHECK_ARG(dz_elems == N * H * W * Cout, "dz does not hold N x H x W x Cout elements");
    SSAD_CHECK_ARG(Cin % 64 == 0 && Cout % 64 == 0, "channel counts must be multiples of 64");
    SSAD_CHECK_ARG((int64_t)Hx * Wx * Cin < (int64_t)1 << 28 && (int64_t)H * W * Cout < (int64_t)1 << 28 &&
                   N * (int64_t)((H + 3) / 4) * ((W + 7) / 8) < (int64_t)1 << 31, "an image must stay below 1 GB (32-bit buffer offsets), tile numbers int");
    static const int s2_tile = getenv("SSAD_WGRAD_HALO_S2_TILE") ? atoi(getenv("SSAD_WGRAD_HALO_S2_TILE")) : 32;
    const bool half = S == 2 && s2_tile == 32;             // stride 2: 4 x 8 output pixels, so that two workgroups fit a CU
    const int TW = half ? 8 : W > 8 ? 16 : 8, TH = half ? 4 : W > 8 ? 4 : 8;     // 64-pixel tiles: 144 accumulator + 44 staging registers fit
    WgHaloParams p;
    p.dz = dz; p.x = x; p.slab = slab; p.N = (int)N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Hx = Hx; p.Wx = Wx;
    p.tiles_y = (H + TH - 1) / TH; p.tiles_x = (W + TW - 1) / TW;
    p.ci_tiles = Cin / 64; p.npairs = (Cin / 64) * (Cout / 64);
    p.ntiles = N * p.tiles_y * p.tiles_x;
    SSAD_CHECK_ARG(splits >= 1 && splits == halo_splits(N * ((H + (W > 8 ? 4 : 8) - 1) / (W > 8 ? 4 : 8)) * ((W + (W > 8 ? 16 : 8) - 1) / (W > 8 ? 16 : 8)), p.npairs),
                   "splits must come from ssad_wgrad3x3_halo_splits");
    p.splits = splits;
    p.chunk = (p.ntiles + splits - 1) / splits;
    const unsigned grid = (unsigned)(p.npairs * splits);
    hipStream_t st = (hipStream_t)stream;
    if (S == 1 && TW == 16) {
        constexpr int bytes = (4 * 16 + 6 * 18) * 64 * 4;
        hipLaunchKernelGGL((wgrad3x3_halo_kernel<4, 16>), dim3(grid), dim3(256), bytes, st, p);
    } else if (S == 1) {
        constexpr int bytes = (8 * 8 + 10 * 10) * 64 * 4;
        hipLaunchKernelGGL((wgrad3x3_halo_kernel<8, 8>), dim3(grid), dim3(256), bytes, st, p);
    } else if (half) {
        constexpr int bytes = (4 * 8 + 9 * 17) * 64 * 4;
        hipLaunchKernelGGL((wgrad3x3_halo_kernel<4, 8, 2>), dim3(grid), dim3(256), bytes, st, p);
    } else if (TW == 16) {
        constexpr int bytes = (4 * 16 + 9 * 33) * 64 * 4;
        static bool set = false;
        if (!set) { SSAD_SET_DYN_LDS((wgrad3x3_halo_kernel<4, 16, 2>), bytes); set = true; }
        hipLaunchKernelGGL((wgrad3x3_halo_kernel<4, 16, 2>), dim3(grid), dim3(256), bytes, st, p);
    } else {
        constexpr int bytes = (8 * 8 + 17 * 17) * 64 * 4;
        static bool set = false;
        if (!set) { SSAD_SET_DYN_LDS((wgrad3x3_halo_kernel<8, 8, 2>), bytes); set = true; }
        hipLaunchKernelGGL((wgrad3x3_halo_kernel<8, 8, 2>), dim3(grid), dim3(256), bytes, st, p);
    }
    SSAD_CHECK_LAUNCH();
    return 0;
}

// dz NHWC [N][H][W][Cout], x NHWC [N][H][W][Cin] (3x3, stride 1, pad 1) -> slab[splits][Cout][9 * Cin] with splits =
// ssad_wgrad3x3_halo_splits(...); follow with ssad_wgrad_reduce(slab, dw, splits, Cout, 9 * Cin, 3, 3, Cin, ...).
extern "C" int ssad_conv_wgrad3x3_halo(const float* dz, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin,
                                       int Cout, int64_t dz_elems, void* stream) {
    return halo_launch(dz, x, slab, splits, N, H, W, H, W, Cin, Cout, 1, dz_elems, stream);
}

// The stride-2 form (ssad_wgrad3x3_halo_ok() == 2): dz NHWC [N][Ho][Wo][Cout] with Ho = (H - 1) / 2 + 1, x NHWC [N][H][W][Cin];
// splits = ssad_wgrad3x3_halo_splits(N, Ho, Wo, Cin, Cout).
extern "C" int ssad_conv_wgrad3x3s2_halo(const float* dz, const float* x, float* slab, int splits, int64_t N, int Ho, int Wo, int H,
                                         int W, int Cin, int Cout, int64_t dz_elems, void* stream) {
    SSAD_CHECK_ARG(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, "dz / x sizes disagree for stride 2, pad 1");
    return halo_launch(dz, x, slab, splits, N, Ho, Wo, H, W, Cin, Cout, 2, dz_elems, stream);
}
