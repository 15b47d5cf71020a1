// 3x3 / stride 1 / pad 1 convolution, register-fed form: the filter goes to the matrix cores from REGISTERS, the halo is staged by extra
// waves of the workgroup.  Two instantiations of one kernel laid out in bytes (16-byte pieces, 128-byte chunk rows, 1 KB fragments):
//   T = hf    the precision-16 step's half tensors (v_mfma_f32_32x32x16_f16): same contract as csrc/conv16.hip -- forward of the
//             torchvision BasicBlock conv3x3 layers under pl.Trainer(precision=16) (models.py:224 / tools.py:263 of the reference) and,
//             with the flipped filter, their input gradients -- for the launches that fill the chip;
//   T = float the exact-fp32 step (v_mfma_f32_32x32x2_f32, four per fragment), replacing csrc/conv_c64.hip / the implicit GEMM on those
//             launches; an inference epilogue (folded BatchNorm) exists and is off (see ssad_conv3x3_fw_eval).
//
// Why (measured on conv16.hip, profiles/r05_conv16_ablate.txt, 256 x 32 x 32 x 128 -> 128: 97 us against a 31 us matrix floor): with
// 64 x 64 wave tiles every v_mfma_f32_32x32x16_f16 needs 1 KB of fragments from LDS -- 128 B/clk/CU at the matrix rate, all the LDS has --
// so the weight slices that go global -> registers -> LDS -> registers (38 us of the 97), the halo writes and the per-tap barriers ADD to
// the matrix stream instead of hiding under it.  Here:
//   * a wave owns 128 pixels x 64 output channels (8 accumulator tiles, 128 registers): 4 activation fragments from LDS and 2 filter
//     fragments per 8 MFMAs -- the LDS serves 64 B/clk;
//   * the filter fragments never touch LDS: the filters are packed once per step (ssad_conv3x3_hw_pack_batch, from the fp32 master
//     weights) in fragment order, [Cout/32][tap][Cin/16][k half][32 channels][8 halves], so a wave's fragment is one coalesced 1 KB
//     read of the L2, requested DB steps ahead into a register ring (sched_barrier per step: the compiler would sink the loads to
//     their uses); no barrier per tap;
//   * the halo is staged by FOUR EXTRA WAVES of the workgroup, one beside each matrix wave (one (tile, chunk) ahead in LDS, loads
//     requested two fills ahead, double-buffered): vmcnt counts in order per wave, so an HBM-latency halo load in a matrix wave's queue
//     would hold back every filter fragment behind it -- the stagers keep their own queue, apply the producer's BatchNorm + ReLU on load
//     (fp32, rounded once) and emit the normalised activation for the weight gradient; all their per-lane index arithmetic is done once;
//   * a residual is ADDED BY THE MATRIX CORES (centre-tap steps against a one-hot fragment: exact), its tiles riding the same staging;
//   * one barrier per (tile, chunk) = per 36 steps.
// Workgroups are persistent over tiles of ONE channel slab (blockIdx.y), so the BatchNorm statistics of the stored output stay in
// registers (per tile in fp32, across tiles in double; the float form: double per value) and leave as one partial row per workgroup.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#ifndef CONV16W_ABL      // timing ablations (tools/micro): 1 = no weight loads, 2 = no halo loads, 4 = no epilogue, 8 = no MFMAs,
                         // 16 = no activation fragment reads, 64 = the epilogue without its stores
#define CONV16W_ABL 0
#endif

namespace {

// T = hf: the precision-16 step's half tensors (v_mfma_f32_32x32x16_f16).  T = float: the exact-fp32 step (v_mfma_f32_32x32x2_f32, four
// per 16-byte fragment: lane half h contracts channels 4 h + q of a group of eight at sub-step q).  Everything is laid out in BYTES --
// 16-byte pieces, 128-byte (or 64-byte) chunk rows, 1 KB fragments -- so both types share the staging, the LDS image and the filter pack.
template <typename T>
struct HWParams {
    const T* in;             // [N][H][W][Cin]
    const T* wp;             // packed filters (ssad_conv3x3_hw_pack_batch / ssad_conv3x3_fw_pack_batch)
    T* out;                  // [N][H][W][Cout]
    const T* residual;       // optional [N][H][W][Cout], added before the rounding
    const uint8_t* res_mask; // optional: one byte per channel quad of the residual, bit k = "pass channel 4 q + k"
    const float* tr_mean;    // optional input transform x <- relu((x - mean) * invstd * gamma + beta), per input channel
    const float* tr_invstd;
    const float* tr_gamma;
    const float* tr_beta;
    const float* shift;      // optional (float): INFERENCE epilogue out = act(acc + shift[c]) -- the folded BatchNorm's shift, its scale
                             // folded into the packed filter (ssad_conv3x3_fw_pack_scaled); no statistics
    int relu;                // ... with act = ReLU
    int64_t os_n;            // (float) element strides of the output / residual tensors: image, row, pixel (NHWC: H W C, W C, C;
    int os_y, os_x;          //  position-major [H][W][N][C]: C, W N C, N C)
    T* emit;                 // optional (with a transform): the transformed input, written once (by channel slab 0)
    double* stats;           // optional [gridDim.x][2][Cout]
    int N, H, W, Cin, Cout;
    int tiles_y, tiles_x, nchunks;
    int64_t ntiles;          // TW16: N * tiles_y * tiles_x;  TW8: ceil(N / 4)
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one fragment deep: a 16-channel MFMA over halves, or four 2-channel MFMAs over floats (element q of both fragments = sub-step q)
__device__ __forceinline__ f32x16 mma_frag(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mma_frag(f32x4 a, f32x4 b, f32x16 c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], c, 0, 0, 0);
    return c;
}

// WN: 64-channel wave columns per workgroup (output channels per workgroup = 64 WN); four matrix waves (WM x WN, WM = 4 / WN) of
// 128 pixels x 64 channels each, four stager waves.  Tile: WN = 2: a 16 x 16 block (18 x 18 halo), or -- TW8, 8 x 8 maps -- four
// consecutive images, each with its 10 x 10 halo; WN = 1 (the 64-channel layer): 16 rows x 32 columns (18 x 34 halo).
// CK: input channels per chunk (64; 32 where two halo stages of 64 channels would not fit the LDS).
// TWP (WN = 1): a tile is TWO consecutive 16 x 16 maps, each with its 18 x 18 halo (the 16 x 16 x 64 maps of the patch-scoring pass).
template <typename T, int WN, bool TW8, int CK, bool TWP = false>
__global__ __launch_bounds__(512) void conv3x3_hw_kernel(HWParams<T> p) {
    constexpr bool F32 = std::is_same<T, float>::value;
    constexpr int E = 16 / (int)sizeof(T);             // elements per 16-byte piece (8 halves / 4 floats)
    constexpr int KST = 2 * E;                         // input channels per step (one fragment deep)
    constexpr int DB = F32 ? 3 : 6;                    // filter fragments are requested DB steps ahead (an fp32 step is 32 MFMAs of 64 cycles)
    using frag_t = typename std::conditional<F32, f32x4, f16x8>::type;
    constexpr int WM = 4 / WN;
    constexpr int TWX = (WN == 1 && !TWP) ? 32 : 16;   // tile width in pixels (not TW8)
    constexpr int LDP = CK + E;                        // elements per LDS halo row (16-byte reads of 16 consecutive pixels: no conflicts)
    constexpr int PPR = CK / E;                        // 16-byte pieces per halo pixel
    constexpr int KS = CK / KST;                       // steps (one fragment deep) per tap
    constexpr int SPC = 9 * KS;                        // steps per chunk
    constexpr int HW_ = TW8 ? 10 : TWX + 2;
    constexpr int NHP = TW8 ? 400 : (TWP ? 2 : 1) * 18 * HW_;
    constexpr int NMW = 4;
    constexpr int SL = 256;                            // stager lanes: one stager wave beside each matrix wave (their instructions
                                                       // delay that wave, and the slowest wave sets the pace at the barrier)
    constexpr int NSP = TW8 ? 256 : NHP;               // staged pixels: an 8 x 8 map's halo ring is all padding, zeroed once
    constexpr int NR = (NSP * PPR + SL - 1) / SL;      // pieces per stager lane per chunk
    constexpr int HALO_H = NHP * LDP;
    constexpr int NT = 512;
    static_assert(SPC % DB == 0 && SPC % 2 == 0, "ring / double-buffer periods must divide a chunk");
    static_assert(!(TW8 && WN == 1), "8 x 8 maps: 128 output channels per workgroup");
    static_assert(!TWP || (WN == 1 && !TW8), "two-map tiles: the 64-channel form");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    T* halo = (T*)lds;                                 // [2][NHP][LDP]
    float* trp = (float*)(halo + 2 * HALO_H);          // [4][Cin]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t my_tiles = p.ntiles > (int64_t)blockIdx.x ? (p.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    // fills of a tile: its nchunks input chunks, then -- with a residual -- the RF chunks of the residual's channel slab: the residual is
    // ADDED BY THE MATRIX CORES (a centre-tap step against a one-hot fragment: fp16 x 1.0 into the fp32 accumulator, exact), so it rides
    // the staging pipeline and its HBM latency is hidden like the input's (first form: 2-byte loads in the epilogue, +20 .. +150 us)
    constexpr int RF = 64 * WN / CK;
    const int fpt = p.nchunks + (p.residual ? RF : 0);
    const int64_t nfill = my_tiles * fpt;
    const int tpi = p.tiles_y * p.tiles_x;

    if (TW8) {                                         // padding rings of both halo stages
        for (int i = tid; i < 2 * HALO_H / E; i += NT) ((u32x4*)halo)[i] = u32x4{0u, 0u, 0u, 0u};
    }
    if (p.tr_mean) {
        for (int c = tid; c < p.Cin; c += NT) {
            trp[c] = p.tr_mean[c]; trp[p.Cin + c] = p.tr_invstd[c]; trp[2 * p.Cin + c] = p.tr_gamma[c]; trp[3 * p.Cin + c] = p.tr_beta[c];
        }
    }
    __syncthreads();

    // The matrix waves issue ahead of the stagers they share a SIMD with: a stager instruction taken first delays the next MFMA by its
    // issue slot, and a fill's worth of them (1-2 us of stager work per fill with everything else ablated) showed up in full in the
    // matrix stream.  (The guard must be provably wave-uniform: s_setprio ignores EXEC.)  Measured: zero-operand launches 83 / 64 / 58 / 55 ->
    // 79 / 61 / 56 / 53 us (halves), 611 -> 600 us (float, 64-channel layer); the steps themselves: unchanged within noise.
#ifndef CONV16W_NO_PRIO
    if (__builtin_amdgcn_readfirstlane(tid) < NMW * 64) __builtin_amdgcn_s_setprio(3);
#endif

    if (wave >= NMW) {
        // =====================================================================================================================
        // stager waves: fill f -> halo[f & 1], one fill ahead of the matrix waves; the loads of fill f + 2 are requested BEFORE the
        // barrier that ends fill f (they have a whole fill of matrix work to arrive: HBM latency is never on the barrier's path)
        // =====================================================================================================================
        // The stagers share two SIMDs with matrix waves: every instruction here delays a matrix wave, and the slowest wave sets the
        // pace at the barrier (first form: ~1 500 instructions of index arithmetic per fill = 1.3 us per fill on every layer).
        // Everything that depends only on (lane, piece) is computed ONCE: byte offset from the halo's first pixel, LDS byte offset,
        // halo coordinates, interior flag; per fill there remain the in-image tests and the uniform base address.
        const int sl = (wave - NMW) * 64 + lane;
        const int piece = sl % PPR;                         // SL is a multiple of PPR: a lane stages the same piece of every pixel
        u32x4 reg[2][NR];                                   // two sets: a fill's loads are requested TWO fills before they are written
        unsigned goff[NR];                                  // bytes from the halo's pixel (-1, -1) (TW8: from image 4 tile, pixel (0, 0))
        unsigned loff[NR];                                  // bytes from the start of a halo stage
        unsigned goffr[NR];                                 // the same in the residual tensor (Cout channels per pixel)
        unsigned hyx[NR];                                   // hy << 8 | hx (TW8: image within the tile)
        unsigned inner = 0, valid = 0, inm[2] = {0u, 0u};
        unsigned rmask[2][NR];                              // the residual's pass bits of each piece: 4 bits (float), 2 x 4 in two bytes (half)
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int hp = (q * SL + sl) / PPR;             // TW8: hp = 64 image + 8 y + x (interior pixels only)
            if (hp < NSP) valid |= 1u << q;
            if (TW8) {
                goff[q] = (unsigned)((hp * p.Cin + piece * E) * sizeof(T));
                goffr[q] = (unsigned)((hp * p.Cout + piece * E) * sizeof(T));
                loff[q] = (unsigned)((((hp >> 6) * 100 + (((hp >> 3) & 7) + 1) * 10 + (hp & 7) + 1) * LDP + piece * E) * sizeof(T));
                hyx[q] = (unsigned)(hp >> 6);
                inner |= 1u << q;
            } else {
                const int img = TWP ? hp / (18 * HW_) : 0, hq = hp - img * 18 * HW_;
                const int hy = hq / HW_, hx = hq - HW_ * hy;
                goff[q] = (unsigned)((((img * p.H + hy) * p.W + hx) * p.Cin + piece * E) * sizeof(T));
                goffr[q] = (unsigned)((((img * p.H + hy) * p.W + hx) * p.Cout + piece * E) * sizeof(T));
                loff[q] = (unsigned)((hp * LDP + piece * E) * sizeof(T));
                hyx[q] = (unsigned)(img << 16 | hy << 8 | hx);
                if (hy >= 1 && hy <= 16 && hx >= 1 && hx <= TWX) inner |= 1u << q;
            }
        }
        // uniform per fill: element index of the halo's pixel (-1, -1) at channel chunk * CK (may lie before the tensor: only
        // in-image pixels are dereferenced), and the tile's position
        // chunk >= nchunks: a residual fill (chunk - nchunks of the workgroup's channel slab), base then indexes the residual tensor
        auto origin = [&](int64_t f, int64_t& base, int& chunk, int& y0, int& x0, int& nimg) {
            const int64_t ti = f / fpt;
            const int k = (int)(f - ti * fpt);
            // order of a tile's fills: input chunk 0, residual chunk 0, input 1, residual 1, ... -- a residual fill is four steps of matrix
            // work, and behind a long input fill its loads (requested two fills ahead) have time to arrive; whatever one kind has more of
            // follows (first form: all residual fills after the last input chunk, +59 us on the fp32 layer2 shape)
            const int mpair = p.residual ? (p.nchunks < RF ? p.nchunks : RF) : 0;
            if (k < 2 * mpair) chunk = (k & 1) ? p.nchunks + (k >> 1) : (k >> 1);
            else chunk = (p.nchunks > mpair ? 0 : p.nchunks) + (k - mpair);
            const int64_t tile = (int64_t)blockIdx.x + ti * gridDim.x;
            const bool res = chunk >= p.nchunks;
            const int C = res ? p.Cout : p.Cin;
            const int c0 = res ? (int)blockIdx.y * (64 * WN) + (chunk - p.nchunks) * CK : chunk * CK;
            if (TW8) {
                y0 = x0 = 0;
                nimg = p.N - (int)(4 * tile);               // images of this tile that exist
                base = (int64_t)(4 * tile) * 64 * C + c0;
            } else if (TWP) {
                y0 = x0 = 0;
                nimg = p.N - (int)(2 * tile);
                base = (((int64_t)(2 * tile) * p.H - 1) * p.W - 1) * C + c0;
            } else {
                const int n0 = (int)(tile / tpi);
                const int rem = (int)(tile - (int64_t)n0 * tpi);
                y0 = (rem / p.tiles_x) * 16;
                x0 = (rem % p.tiles_x) * TWX;
                nimg = 1;
                base = (((int64_t)n0 * p.H + y0 - 1) * p.W + x0 - 1) * C + c0;
            }
        };
        auto load_fill = [&](int64_t f, auto set_tag) {
            constexpr int SET = decltype(set_tag)::value;
            int64_t base; int chunk, y0, x0, nimg;
            origin(f, base, chunk, y0, x0, nimg);
            const bool res = chunk >= p.nchunks;
            const char* src = (const char*)((res ? p.residual : p.in) + base);
            unsigned im = 0;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                bool ok = (valid >> q) & 1u;
                if (res) ok = ok && ((inner >> q) & 1u);   // only the centre tap reads a residual fill
                if (TW8) ok = ok && (int)hyx[q] < nimg;
                else ok = ok && (unsigned)(y0 - 1 + (int)((hyx[q] >> 8) & 255u)) < (unsigned)p.H && (unsigned)(x0 - 1 + (int)(hyx[q] & 255u)) < (unsigned)p.W
                          && (int)(hyx[q] >> 16) < nimg;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (ok && !(CONV16W_ABL & 2)) v = *(const u32x4*)(src + (res ? goffr[q] : goff[q]));
                reg[SET][q] = v;
                {
                    // the identity-branch gradient is (dy, nibble mask), never materialised: one mask byte per channel quad, i.e. per
                    // 16-byte piece one byte (float) or two (half: 8 channels), at the piece's element index / 4
                    unsigned mk = 0xffffu;
                    if (res && p.res_mask && ok) {
                        if (F32) mk = p.res_mask[(base * (int64_t)sizeof(T) + goffr[q]) >> 4];
                        else mk = *(const uint16_t*)(p.res_mask + ((base * (int64_t)sizeof(T) + goffr[q]) >> 3));
                    }
                    rmask[SET][q] = mk;
                }
                im |= (ok ? 1u : 0u) << q;
            }
            inm[SET] = im;
        };
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        auto write_fill = [&](int64_t f, auto set_tag) {
            constexpr int SET = decltype(set_tag)::value;
            int64_t base; int chunk, y0, x0, nimg;
            origin(f, base, chunk, y0, x0, nimg);
            char* dst = (char*)(halo + (int)(f & 1) * HALO_H);
            if (p.tr_mean && chunk < p.nchunks) {
                // producer's train-mode BatchNorm + ReLU on load: bn_apply_fwd's expression in fp32 (two channels per packed
                // instruction); halves: rounded once.  Zero padding pads the TRANSFORMED activation: out-of-image pieces stay zero.
                // The E channels of this lane's piece: read once per fill.
                const int c = chunk * CK + piece * E;
                f32x2 mu[E / 2], sc[E / 2], ga[E / 2], be[E / 2];
#pragma unroll
                for (int k = 0; k < E / 2; ++k) {
                    mu[k] = *(const f32x2*)(trp + c + 2 * k); sc[k] = *(const f32x2*)(trp + p.Cin + c + 2 * k);
                    ga[k] = *(const f32x2*)(trp + 2 * p.Cin + c + 2 * k); be[k] = *(const f32x2*)(trp + 3 * p.Cin + c + 2 * k);
                }
                char* em = (p.emit && blockIdx.y == 0) ? (char*)(p.emit + base) : nullptr;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    if (!((inm[SET] >> q) & 1u)) continue;
                    // (written over an element vector: with the piece held as four dwords and the pairs bit-cast out of / into its
                    // elements, hipcc fed pair 0's RESULT to pairs 1-3 -- found by the emitted activation, tests/test_hip_half.py)
                    const frag_t v = __builtin_bit_cast(frag_t, reg[SET][q]);
                    frag_t o;
#pragma unroll
                    for (int k = 0; k < E / 2; ++k) {
                        const f32x2 xf = {(float)v[2 * k], (float)v[2 * k + 1]};
                        const f32x2 y = (xf - mu[k]) * sc[k] * ga[k] + be[k];
                        o[2 * k] = (T)fmaxf(y[0], 0.f);
                        o[2 * k + 1] = (T)fmaxf(y[1], 0.f);
                    }
                    const u32x4 w = __builtin_bit_cast(u32x4, o);
                    reg[SET][q] = w;
                    // interior pixels of the halo: the activation this layer's weight gradient reads
                    if (em && ((inner >> q) & 1u)) *(u32x4*)(em + goff[q]) = w;
                }
            }
            if (p.res_mask && chunk >= p.nchunks) {
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    u32x4 w = reg[SET][q];
                    const unsigned mk = rmask[SET][q];
                    if (F32) {
                        w[0] = (mk & 1u) ? w[0] : 0u; w[1] = (mk & 2u) ? w[1] : 0u; w[2] = (mk & 4u) ? w[2] : 0u; w[3] = (mk & 8u) ? w[3] : 0u;
                    } else {
                        // dword j holds channels 2 j (low half) and 2 j + 1: bits 0-3 of the first byte, bits 0-3 of the second
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned b0 = j < 2 ? 2 * j : 8 + 2 * (j - 2);
                            const unsigned keep = ((mk >> b0) & 1u ? 0x0000ffffu : 0u) | ((mk >> (b0 + 1)) & 1u ? 0xffff0000u : 0u);
                            w[j] &= keep;
                        }
                    }
                    reg[SET][q] = w;
                }
            }
#pragma unroll
            for (int q = 0; q < NR; ++q)
                if ((valid >> q) & 1u) *(u32x4*)(dst + loff[q]) = reg[SET][q];
        };
        // fill g is loaded into set g & 1 while fill g - 2 (same set, already written) is being consumed: HBM has two fills of matrix
        // work to answer (one was not enough on the 64-channel layer: 18 steps per fill)
        const std::integral_constant<int, 0> S0;
        const std::integral_constant<int, 1> S1;
        if (nfill > 0) { load_fill(0, S0); write_fill(0, S0); }
        if (nfill > 1) load_fill(1, S1);
        if (nfill > 2) load_fill(2, S0);
        __syncthreads();
        for (int64_t f = 0; f < nfill; f += 2) {
            if (f + 1 < nfill) write_fill(f + 1, S1);
            if (f + 3 < nfill) load_fill(f + 3, S1);
            __syncthreads();
            if (f + 1 >= nfill) break;
            if (f + 2 < nfill) write_fill(f + 2, S0);
            if (f + 4 < nfill) load_fill(f + 4, S0);
            __syncthreads();
        }
        if (p.stats) __syncthreads();
        return;
    }

    // =========================================================================================================================
    // matrix waves
    // =========================================================================================================================
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int co0 = blockIdx.y * (64 * WN) + wn * 64;          // first output channel of this wave
    // rows / columns of the tile this wave's 128 pixels start at (TW8: the wave's first image)
    const int wrow = TW8 ? 0 : (WN == 1 ? 8 * (wm >> 1) : 8 * wm), wcol = (!TW8 && WN == 1 && !TWP) ? 16 * (wm & 1) : 0;
    const int wimg = TWP ? (wm & 1) : 0;                       // TWP: the wave's map of the tile's two

    int abase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (TW8) abase[m] = ((2 * wm + (m >> 1)) * 100 + (4 * (m & 1) + (r >> 3)) * 10 + (r & 7)) * LDP + E * h;
        else abase[m] = (wimg * 18 * HW_ + (wrow + 2 * m + (r >> 4)) * HW_ + wcol + (r & 15)) * LDP + E * h;
    }
    const int KB = p.Cin / KST;                                // fragments (64 lanes x 16 bytes = 64 E elements) per (channel tile, tap)
    const T* bptr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bptr[j] = p.wp + (int64_t)(co0 / 32 + j) * 9 * KB * (64 * E) + lane * E;

    // Every workgroup walks the nine taps in its own rotation (conv16.hip: persistent workgroups run in lockstep, with one common order
    // all of them ask the L2 for the same lines at the same moment).  fp32 accumulation order differs between workgroups by the
    // rotation only: deterministic for a given launch geometry.
    // (fp32: no rotation -- a fragment is requested every 2 048 cycles, and one common tap order keeps every output value's summation
    // order independent of the launch geometry)
    const int rot = F32 ? 0 : (int)((blockIdx.x + blockIdx.y) % 9);
    int toffA[9], tapB[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int te = t + rot;
        te = te >= 9 ? te - 9 : te;
        const int ty3 = te >= 6 ? 2 : te >= 3 ? 1 : 0;
        toffA[t] = (ty3 * HW_ + (te - 3 * ty3)) * LDP;
        tapB[t] = te * KB * (64 * E);
    }

    f32x16 acc[4][2];
    double st0[2] = {0.0, 0.0}, st1[2] = {0.0, 0.0};

    frag_t breg[DB][2];
    auto load_b = [&](int chunk, int s, int set) {        // s: step within a chunk (compile-time after unrolling)
        if (CONV16W_ABL & 1) return;
        const int off = tapB[s / KS] + (chunk * KS + (s % KS)) * (64 * E);
#pragma unroll
        for (int j = 0; j < 2; ++j) breg[set][j] = *(const frag_t*)(bptr[j] + off);
    };
    if (CONV16W_ABL & 1) {
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < E; ++k) breg[d][j][k] = (T)0.f;
    }
    if (nfill > 0) {
#pragma unroll
        for (int d = 0; d < DB; ++d) load_b(0, d, d);
    }
    // this lane's share of an output address, in bytes: pixel column 4 h of the wave's first row, channel r of the wave's first 32
    const unsigned lane_off = F32 ? (unsigned)((wrow * p.os_y + (wcol + 4 * h) * p.os_x + wn * 64 + r) * sizeof(T))
                                  : (unsigned)(((wrow * p.W + wcol + 4 * h) * p.Cout + wn * 64 + r) * sizeof(T));
    typedef hf h2 __attribute__((ext_vector_type(2)));
    __syncthreads();                                       // fill 0 is staged

    int chunk = 0;
    int64_t tile_i = 0;
    int64_t f = 0;                                         // fills consumed so far (halo stage = f & 1)
    const int64_t nin = my_tiles * p.nchunks;              // input fills
    for (int64_t fi = 0; fi < nin; ++fi) {
        const T* hb = halo + (int)(f & 1) * HALO_H;
        const int cnext = chunk + 1 == p.nchunks ? 0 : chunk + 1;
        const bool more = cnext != 0 || tile_i + 1 < my_tiles;          // another input fill follows (this tile's or the next tile's)
        frag_t areg[2][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) areg[0][m] = *(const frag_t*)(hb + abase[m] + toffA[0]);
        if (CONV16W_ABL & 16) {
#pragma unroll
            for (int m = 0; m < 4; ++m) areg[1][m] = areg[0][m];
        }
#pragma unroll
        for (int s = 0; s < SPC; ++s) {
            if (s + 1 < SPC && !(CONV16W_ABL & 16)) {
#pragma unroll
                for (int m = 0; m < 4; ++m) areg[(s + 1) & 1][m] = *(const frag_t*)(hb + abase[m] + toffA[(s + 1) / KS] + ((s + 1) % KS) * KST);
            }
            __builtin_amdgcn_sched_barrier(0);             // the next step's activation fragments are requested BEFORE this step's MFMAs
            if (s == 0 && chunk == 0) {                    // first step of a tile: accumulate onto zero (no register clearing)
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[m][j] = mma_frag(areg[0][m], breg[0][j], zero);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (!(CONV16W_ABL & 8)) acc[m][j] = mma_frag(areg[s & 1][m], breg[s % DB][j], acc[m][j]);
            }
            // the register set just consumed takes the fragments of the step DB ahead (this chunk, the next chunk, or the first
            // chunk of the next tile: the same filters)
            if (s + DB < SPC) load_b(chunk, s + DB, s % DB);
            else if (more) load_b(cnext, s + DB - SPC, s % DB);
            // the machine scheduler may not move anything across a step: left alone it sinks every fragment load to its use (fewer
            // live registers) and the matrix stream then waits for the L2 each step -- the software pipeline IS the kernel
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                   // everyone is done with halo[f & 1]; fill f + 1 is staged
        ++f;
        // ---- a residual fill: acc += residual x one-hot.  Step kk of fill rc covers the slab's channels rc CK + KST kk .. + KST - 1; a wave
        // takes the steps inside its own 64 channels: fragment element j of lane (r, h) of accumulator tile jt is 1 where
        // jt 32 + r = cb + E h + j (halves: the tile a step does not belong to adds zeros).  Residual fill rc follows input chunk rc
        // (see the stagers); those beyond the number of input chunks follow the last one ----
        auto residual_fill = [&](int rc) {
            const T* hr = halo + (int)(f & 1) * HALO_H;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int cb = rc * CK + kk * KST - wn * 64;          // first channel of the step, counted from the wave's first
                if (cb >= 0 && cb < 64) {
                    frag_t ar[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) ar[m] = *(const frag_t*)(hr + abase[m] + (HW_ + 1) * LDP + kk * KST);
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt) {
                        if (F32 && jt != (cb >> 5)) continue;         // (float: a step lies inside one accumulator tile; 4 MFMAs saved per skip)
                        const int hot = jt * 32 + r - cb - E * h;
                        frag_t one;
#pragma unroll
                        for (int j = 0; j < E; ++j) one[j] = hot == j ? (T)1.f : (T)0.f;
#pragma unroll
                        for (int m = 0; m < 4; ++m) acc[m][jt] = mma_frag(ar[m], one, acc[m][jt]);
                    }
                }
            }
            __syncthreads();
            ++f;
        };
        if (p.residual && chunk < RF) residual_fill(chunk);
        const int done = chunk;
        chunk = cnext;
        if (chunk != 0) continue;
        if (p.residual) {
            for (int rc = done + 1; rc < RF; ++rc) residual_fill(rc);
        }

        // ---- epilogue of a finished tile: straight from the accumulators, every lane its own halves (a wave store covers whole
        // 64-byte runs: 32 consecutive channels of one pixel per lane half).  Addresses are a uniform base per (tile, register) plus
        // this lane's constant byte offset; two registers (neighbouring pixels of one channel) are rounded as a pair, the BatchNorm
        // statistics of the stored halves are two v_dot2_f32_f16 per pair (products of halves are exact in fp32) ----
        const int64_t tile = (int64_t)blockIdx.x + tile_i * gridDim.x;
        ++tile_i;
        if (CONV16W_ABL & 128) continue;                   // ablation 128: a tile ends without a single instruction reading its accumulators
        if (CONV16W_ABL & 4) {
            float sum = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sum += acc[m][j][e];
            if (sum == 123.456f) p.out[0] = (T)sum;
            continue;
        }
        int64_t torg;                                      // first element of the tile (TW8: of image 4 tile), this workgroup's slab
        const int rowst = F32 ? p.os_y : p.W * p.Cout;     // (float: output strides are parameters -- NHWC or position-major)
        const int pixst = F32 ? p.os_x : p.Cout;
        const int64_t imgst = F32 ? p.os_n : (int64_t)p.H * p.W * p.Cout;
        bool wave_ok = true;
        if (TW8) {
            torg = (int64_t)(4 * tile) * imgst;
        } else if (TWP) {
            torg = (int64_t)(2 * tile + wimg) * imgst;
            wave_ok = (int)(2 * tile) + wimg < p.N;
        } else {
            const int n0 = (int)(tile / tpi);
            const int rem = (int)(tile - (int64_t)n0 * tpi);
            torg = (int64_t)n0 * imgst + (int64_t)((rem / p.tiles_x) * 16) * rowst + (int64_t)((rem % p.tiles_x) * TWX) * pixst;
        }
        torg += blockIdx.y * (64 * WN);
        float fs[2] = {0.f, 0.f}, fq[2] = {0.f, 0.f};
        const h2 ones = {(hf)1.f, (hf)1.f};
        // laundered per tile: the 128 per-register offsets below are loop invariants, and hoisted out of the tile loop they are kept
        // alive (in scratch: 119 spilled registers) through the matrix loop
        unsigned lo = lane_off;
        asm volatile("" : "+v"(lo));
        // (the statistics are always taken -- two instructions per pair; a per-register branch on p.stats made the compiler park the
        // rounded pairs in scratch between the store and the statistics blocks)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            // pixel of register e (lane half h): TW16: row wrow + 2 m + (e >> 3), column wcol + (e & 3) + 4 h + 8 ((e >> 2) & 1);
            //                                    TW8: image 2 wm + (m >> 1), row 4 (m & 1) + (e >> 2), column (e & 3) + 4 h
            int64_t moff;
            bool ok = true;
            if (TW8) {
                ok = (int)(4 * tile) + 2 * wm + (m >> 1) < p.N;
                moff = torg + (int64_t)(2 * wm + (m >> 1)) * imgst + (int64_t)(4 * (m & 1)) * rowst;
            } else {
                ok = wave_ok;
                moff = torg + (int64_t)(2 * m) * rowst;
            }
            if (!ok) continue;
            char* const ob = (char*)(p.out + moff);
            if constexpr (F32) {
                // exact-fp32 step: values stored as they are, statistics in double per value (as the other fp32 kernels take them;
                // an fp32 tile is ~300 k cycles of matrix work, its 128 conversions do not show).  Inference (p.shift): the folded
                // BatchNorm's shift and the ReLU instead, no statistics.
                float sh[2] = {0.f, 0.f};
                if (p.shift) { sh[0] = p.shift[co0 + r]; sh[1] = p.shift[co0 + 32 + r]; }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int eo = TW8 ? (e >> 2) * rowst + (e & 3) * pixst : (e >> 3) * rowst + ((e & 3) + 8 * ((e >> 2) & 1)) * pixst;
                    const unsigned o0 = lo + 4u * (unsigned)eo;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float v = acc[m][j][e];
                        if (p.shift) {
                            v += sh[j];
                            if (p.relu) v = fmaxf(v, 0.f);
                        } else {
                            st0[j] += (double)v;
                            st1[j] += (double)v * (double)v;
                        }
                        *(float*)(ob + o0 + 128 * j) = v;
                    }
                }
            } else {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const int eo = TW8 ? (e >> 2) * rowst + (e & 3) * p.Cout : (e >> 3) * rowst + ((e & 3) + 8 * ((e >> 2) & 1)) * p.Cout;
                const unsigned o0 = lo + 2u * (unsigned)eo, o1 = o0 + 2u * (unsigned)p.Cout;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const h2 pk = {(hf)acc[m][j][e], (hf)acc[m][j][e + 1]};
                    if (!(CONV16W_ABL & 64) || fs[j] == 123.456f) {       // ablation 64: the epilogue's arithmetic without its stores
                        *(hf*)(ob + o0 + 64 * j) = pk[0];
                        *(hf*)(ob + o1 + 64 * j) = pk[1];
                    }
                    fs[j] = __builtin_amdgcn_fdot2(pk, ones, fs[j], false);        // statistics of what is stored
                    fq[j] = __builtin_amdgcn_fdot2(pk, pk, fq[j], false);
                }
            }
            }
        }
        // (unconditional: under `if (p.stats)` the compiler sinks the dot products into the conditional block and keeps every pair alive)
#pragma unroll
        for (int j = 0; j < 2; ++j) { st0[j] += (double)fs[j]; st1[j] += (double)fq[j]; }
    }

    if (p.stats) {
        // lane halves, then the row-waves in a fixed order through LDS (both halo stages are dead: the loop ended on a barrier)
        double* S = (double*)lds;                  // [WM][2][64 WN]
        constexpr int BN = 64 * WN;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            st0[j] += __shfl_xor(st0[j], 32);
            st1[j] += __shfl_xor(st1[j], 32);
            if (h == 0) {
                S[(wm * 2 + 0) * BN + wn * 64 + j * 32 + r] = st0[j];
                S[(wm * 2 + 1) * BN + wn * 64 + j * 32 + r] = st1[j];
            }
        }
        __syncthreads();
        for (int u = tid; u < 2 * BN; u += NMW * 64) {
            const int which = u / BN, cc = u % BN;
            double t = S[(0 * 2 + which) * BN + cc];
#pragma unroll
            for (int w = 1; w < WM; ++w) t += S[(w * 2 + which) * BN + cc];
            p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + blockIdx.y * BN + cc] = t;
        }
    }
}

// ---- filters in fragment order: dst[Cout/32][9][Cin/16][2][32][8] <- fp32 OHWI master weights (or their flipped transposes) ----
struct PackTable {
    int n;
    int64_t e[32][6];       // src offset (floats), dst offset (halves), O, I (of the PACKED conv), flip, first block
};

template <typename T>
__global__ __launch_bounds__(256) void pack_hw_kernel(const float* __restrict__ src, T* __restrict__ dst, PackTable t,
                                                      const float* __restrict__ oscale = nullptr) {
    constexpr int E = 16 / (int)sizeof(T), KST = 2 * E;
    using frag_t = typename std::conditional<std::is_same<T, float>::value, f32x4, f16x8>::type;
    int k = 0;
    while (k + 1 < t.n && (int64_t)blockIdx.x >= t.e[k + 1][5]) ++k;
    const int O = (int)t.e[k][2], I = (int)t.e[k][3], flip = (int)t.e[k][4];
    const int KB = I / KST;
    const int64_t piece = ((int64_t)blockIdx.x - t.e[k][5]) * 256 + threadIdx.x;       // 16-byte piece of the destination
    if (piece >= (int64_t)O * 9 * I / E) return;
    const int n = (int)(piece & 31), kh = (int)((piece >> 5) & 1);
    int64_t rest = piece >> 6;
    const int kb = (int)(rest % KB); rest /= KB;
    const int tap = (int)(rest % 9);
    const int ct = (int)(rest / 9);
    const int o = ct * 32 + n, i0 = kb * KST + kh * E;
    const float* w = src + t.e[k][0];
    frag_t v;
    const float osc = oscale ? oscale[o] : 1.f;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        // plain: this conv's OHWI filter [O][9][I].  flip: this conv is the input gradient of a conv whose filter is [I][9][O]:
        // its weight (o, tap, i) is that filter's (i, 8 - tap, o)
        // (one load per element at a selected INDEX, not a load in each arm of the select: hipcc waited for every such load before the
        // next one -- eight L2 round trips in a row per thread, round 6)
        const int64_t src_i = flip ? ((int64_t)(i0 + j) * 9 + (8 - tap)) * O + o : ((int64_t)o * 9 + tap) * I + i0 + j;
        const float x = w[src_i];
        v[j] = (T)(x * osc);                             // inference: the folded BatchNorm's scale of output channel o (else 1: exact)
    }
    *(frag_t*)(dst + t.e[k][1] + piece * E) = v;
}

struct GeoW {
    bool tw8, ck32, twp;
    int tiles_y, tiles_x, wn, gx, gy;
    int64_t ntiles;
};

static GeoW geometry_w(int64_t N, int H, int W, int Cout) {
    GeoW g;
    g.tw8 = H == 8 && W == 8;
    g.wn = Cout % 128 == 0 ? 2 : 1;
    g.ck32 = g.wn == 1;                   // 16 x 32 tiles: two 18 x 34 halo stages of 64 channels would not fit the LDS
    g.twp = g.wn == 1 && H == 16 && W == 16;               // 16 x 16 maps of 64 channels: two maps per tile
    g.tiles_y = g.tw8 ? 1 : H / 16;
    g.tiles_x = g.tw8 ? 1 : W / (g.wn == 1 ? 32 : 16);
    g.ntiles = g.tw8 ? (N + 3) / 4 : g.twp ? (N + 1) / 2 : N * g.tiles_y * g.tiles_x;
    g.gy = Cout / (64 * g.wn);
    static const int slots = getenv("SSAD_CONV16W_WGS") ? atoi(getenv("SSAD_CONV16W_WGS")) : 256;      // one workgroup per CU
    int64_t gx = slots / g.gy;
    if (gx < 1) gx = 1;
    if (gx > g.ntiles) gx = g.ntiles;
    g.gx = (int)gx;
    return g;
}

// bytes of the two halo stages (+ 16 KB: the largest transform table); CKB = bytes of a chunk row (128, or 64 for 16 x 32 tiles)
static constexpr int lds_bytes_w(bool wn1, bool tw8, bool twp = false) {
    return 2 * (tw8 ? 400 : twp ? 648 : 18 * (wn1 ? 34 : 18)) * ((wn1 ? 64 : 128) + 16) + 16 * 1024;
}

static int shape_ok(int64_t N, int H, int W, int Cin, int Cout) {
    if (Cin % 64 || Cout % 64 || Cin > 1024 || N <= 0) return 0;
    const bool wn1 = Cout % 128 != 0;                     // 64 output channels per workgroup: 16 x 32 tiles
    if (!((H == 8 && W == 8 && !wn1) || (wn1 && H == 16 && W == 16) || (H > 0 && W > 0 && H % 16 == 0 && W % (wn1 ? 32 : 16) == 0))) return 0;
    const GeoW g = geometry_w(N, H, W, Cout);
    static const int min_items = getenv("SSAD_CONV16W_MIN") ? atoi(getenv("SSAD_CONV16W_MIN")) : 200;
    return g.ntiles * g.gy >= min_items;
}

template <typename T>
static int pack_batch_impl(const float* src, T* dst, const int64_t* desc, int n, void* stream) {
    constexpr int E = 16 / (int)sizeof(T);
    SSAD_CHECK_ARG(src && dst && desc && n > 0, "bad argument");
    for (int k = 0; k < n; ++k)
        SSAD_CHECK_ARG(desc[5 * k + 2] > 0 && desc[5 * k + 3] > 0 && desc[5 * k + 2] % 64 == 0 && desc[5 * k + 3] % 64 == 0 &&
                       desc[5 * k + 1] % E == 0, "bad filter shape (channel counts multiples of 64)");
    for (int base = 0; base < n; base += 32) {
        PackTable t;
        t.n = n - base < 32 ? n - base : 32;
        int64_t acc = 0;
        for (int k = 0; k < t.n; ++k) {
            for (int j = 0; j < 5; ++j) t.e[k][j] = desc[5 * (base + k) + j];
            t.e[k][5] = acc;
            acc += cdiv64(t.e[k][2] * 9 * t.e[k][3] / E, 256);
        }
        SSAD_CHECK_ARG(acc < (int64_t)2147483647, "too many blocks");
        hipLaunchKernelGGL(pack_hw_kernel<T>, dim3((unsigned)acc), dim3(256), 0, (hipStream_t)stream, src, dst, t);
        SSAD_CHECK_LAUNCH();
    }
    return 0;
}

template <typename T>
static int conv_impl(const T* in, const T* w_packed, T* out, const T* residual, const uint8_t* res_mask, const float* tr_mean,
                     const float* tr_invstd, const float* tr_gamma, const float* tr_beta, T* emit, int64_t N, int H, int W, int Cin,
                     int Cout, double* stats_ws, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                     float* running_var, void* stream, const float* shift = nullptr, int relu = 0, int out_hwnc = 0) {
    constexpr int CKW = 128 / (int)sizeof(T), CKN = 64 / (int)sizeof(T);     // channels per chunk: 128-byte rows; 64-byte rows for 16 x 32 tiles
    SSAD_CHECK_ARG(in && w_packed && out && N > 0 && H > 0 && W > 0, "bad argument");
    SSAD_CHECK_ARG(Cin % 64 == 0 && Cout % 64 == 0 && Cin <= 1024, "channel counts must be multiples of 64 (Cin <= 1024)");
    SSAD_CHECK_ARG((H == 8 && W == 8 && Cout % 128 == 0) || (H == 16 && W == 16) || (H % 16 == 0 && W % (Cout % 128 == 0 ? 16 : 32) == 0),
                   "maps of 16 x 16 blocks (16 x 32 when Cout is not a multiple of 128, or 16 x 16 maps), or 8 x 8 maps with Cout a multiple of 128");
    SSAD_CHECK_ARG(!(shift || out_hwnc) || (std::is_same<T, float>::value && !stats_ws), "inference epilogue / position-major output: float, no statistics");
    SSAD_CHECK_ARG(!tr_mean || (tr_invstd && tr_gamma && tr_beta), "input transform needs mean, invstd, gamma, beta");
    SSAD_CHECK_ARG(!stats_ws || (mean && invstd), "statistics need mean / invstd outputs");
    SSAD_CHECK_ARG(!emit || tr_mean, "emit without an input transform");
    SSAD_CHECK_ARG(!res_mask || residual, "a residual mask without a residual");
    SSAD_CHECK_ARG(N * (int64_t)H * W < (int64_t)1 << 31, "too many pixels for one launch");
    const GeoW g = geometry_w(N, H, W, Cout);
    HWParams<T> p;
    p.in = in; p.wp = w_packed; p.out = out; p.residual = residual; p.res_mask = res_mask;
    p.tr_mean = tr_mean; p.tr_invstd = tr_invstd; p.tr_gamma = tr_gamma; p.tr_beta = tr_beta; p.emit = emit;
    p.stats = stats_ws;
    p.shift = shift; p.relu = relu;
    p.os_n = out_hwnc ? Cout : (int64_t)H * W * Cout; p.os_y = out_hwnc ? (int)(W * N * Cout) : W * Cout; p.os_x = out_hwnc ? (int)(N * Cout) : Cout;
    SSAD_CHECK_ARG(!out_hwnc || (int64_t)H * W * N * Cout < (int64_t)1 << 31, "position-major output too large for 32-bit strides");
    p.N = (int)N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.tiles_y = g.tiles_y; p.tiles_x = g.tiles_x; p.nchunks = Cin / (g.ck32 ? CKN : CKW); p.ntiles = g.ntiles;
    const dim3 grid((unsigned)g.gx, (unsigned)g.gy);
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        SSAD_SET_DYN_LDS((conv3x3_hw_kernel<T, 2, true, CKW>), lds_bytes_w(false, true));
        SSAD_SET_DYN_LDS((conv3x3_hw_kernel<T, 2, false, CKW>), lds_bytes_w(false, false));
        SSAD_SET_DYN_LDS((conv3x3_hw_kernel<T, 1, false, CKN>), lds_bytes_w(true, false));
        SSAD_SET_DYN_LDS((conv3x3_hw_kernel<T, 1, false, CKN, true>), lds_bytes_w(true, false, true));
        attr_set = true;
    }
    const int lds_dyn = lds_bytes_w(g.wn == 1, g.tw8, g.twp) - 16 * 1024 + (tr_mean ? 16 * Cin : 0);
    if (g.wn == 2) {
#ifdef CONV16W_FORCE_CKN      // experiment (tools/micro): the 128-channel form with half-size fills -- twice the barriers, same MFMAs
        if (!g.tw8) {
            p.nchunks = Cin / CKN;
            hipLaunchKernelGGL((conv3x3_hw_kernel<T, 2, false, CKN>), grid, dim3(512), lds_dyn, st, p);
            SSAD_CHECK_LAUNCH();
            return 0;
        }
#endif
        if (g.tw8) hipLaunchKernelGGL((conv3x3_hw_kernel<T, 2, true, CKW>), grid, dim3(512), lds_dyn, st, p);
        else hipLaunchKernelGGL((conv3x3_hw_kernel<T, 2, false, CKW>), grid, dim3(512), lds_dyn, st, p);
    } else if (g.twp) {
        hipLaunchKernelGGL((conv3x3_hw_kernel<T, 1, false, CKN, true>), grid, dim3(512), lds_dyn, st, p);
    } else {
        hipLaunchKernelGGL((conv3x3_hw_kernel<T, 1, false, CKN>), grid, dim3(512), lds_dyn, st, p);
    }
    SSAD_CHECK_LAUNCH();
    if (stats_ws)
        return ssad_bn_finalize_partials(stats_ws, g.gx, N * H * W, Cout, eps, momentum, mean, invstd, running_mean, running_var, stream);
    return 0;
}

}  // namespace

// 1 when ssad_conv3x3_hw takes the launch: channel counts multiples of 64, maps of 16 x 16 blocks (16 x 32 for 64 output channels; or
// 8 x 8 maps), and enough (tile, channel slab) pairs to give every CU a workgroup (smaller launches stay on csrc/conv16.hip).
extern "C" int ssad_conv3x3_hw_ok(int64_t N, int H, int W, int Cin, int Cout) {
    static const int on = getenv("SSAD_CONV16W") ? atoi(getenv("SSAD_CONV16W")) : 1;
    return on && shape_ok(N, H, W, Cin, Cout);
}
// the same for the exact-fp32 form (ssad_conv3x3_fw); smaller launches stay on csrc/conv_c64.hip / csrc/conv_igemm.hip
// (and at least two tiles per workgroup, or tiles of >= 8 chunks: a workgroup's first halo and its last epilogue are not overlapped,
// measured on the 64-channel layer at batch 32 -- one 70 us tile per workgroup -- the batch-32 step lost 0.08 ms against the c64 kernel)
extern "C" int ssad_conv3x3_fw_ok(int64_t N, int H, int W, int Cin, int Cout) {
    static const int on = getenv("SSAD_CONV32W") ? atoi(getenv("SSAD_CONV32W")) : 1;
    if (!on || !shape_ok(N, H, W, Cin, Cout)) return 0;
    const GeoW g = geometry_w(N, H, W, Cout);
    return g.ntiles * g.gy >= 512 || Cin >= 256;
}

// elements of one packed filter of Cout x 3 x 3 x Cin (halves / floats)
extern "C" int64_t ssad_conv3x3_hw_packed_size(int Cin, int Cout) { return (int64_t)Cout * 9 * Cin; }

// rows of the statistics workspace (x 2 x Cout doubles), both forms
extern "C" int64_t ssad_conv3x3_hw_stats_rows(int64_t N, int H, int W, int Cout) { return geometry_w(N, H, W, Cout).gx; }

// Packs n filters in one launch.  desc[5 k ..]: source offset (floats from src), destination offset (elements from dst), Cout, Cin of
// the conv that will RUN on the packed filter, flip (0: src holds that conv's OHWI filter [Cout][3][3][Cin]; 1: src holds the OHWI
// filter [Cin][3][3][Cout] of the forward conv whose input gradient this is).  _hw: values rounded to halves as ssad_cvt_f32_f16 does;
// _fw: floats, fragment order [Cout/32][tap][Cin/8][2][32][4].
extern "C" int ssad_conv3x3_hw_pack_batch(const float* src, void* dst, const int64_t* desc, int n, void* stream) {
    return pack_batch_impl<hf>(src, (hf*)dst, desc, n, stream);
}
extern "C" int ssad_conv3x3_fw_pack_batch(const float* src, float* dst, const int64_t* desc, int n, void* stream) {
    return pack_batch_impl<float>(src, dst, desc, n, stream);
}

// ssad_conv3x3_h (csrc/conv16.hip) with the filter in packed form: out = conv3x3(pad 1, stride 1)(T(in)) (+ residual), half tensors
// NHWC; T = identity or relu((x - tr_mean) * tr_invstd * tr_gamma + tr_beta) per input channel; emit receives T(in); stats_ws != NULL:
// train-mode BatchNorm statistics of the stored output (ssad_conv3x3_hw_stats_rows(...) * 2 * Cout doubles), finalised as
// ssad_conv_igemm_fwd_stats does.  The launch must satisfy ssad_conv3x3_hw_ok.
extern "C" int ssad_conv3x3_hw(const void* in, const void* w_packed, void* out, const void* residual, const uint8_t* res_mask,
                               const float* tr_mean, const float* tr_invstd, const float* tr_gamma, const float* tr_beta, void* emit,
                               int64_t N, int H, int W, int Cin, int Cout, double* stats_ws, float eps, float momentum, float* mean,
                               float* invstd, float* running_mean, float* running_var, void* stream) {
    return conv_impl<hf>((const hf*)in, (const hf*)w_packed, (hf*)out, (const hf*)residual, res_mask, tr_mean, tr_invstd, tr_gamma, tr_beta,
                         (hf*)emit, N, H, W, Cin, Cout, stats_ws, eps, momentum, mean, invstd, running_mean, running_var, stream);
}

// The exact-fp32 form of the same kernel (fp32 tensors, v_mfma_f32_32x32x2_f32, statistics in double per value): the 3 x 3 / stride 1
// convs of the fp32 training step forward and -- with the flipped pack -- their input gradients.  res_mask (optional): the residual is the
// identity-branch gradient (dy, nibble mask) of a residual block, one byte per channel quad as ssad_bn_apply_fwd_mask writes them.
extern "C" int ssad_conv3x3_fw(const float* in, const float* w_packed, float* out, const float* residual, const uint8_t* res_mask,
                               const float* tr_mean, const float* tr_invstd, const float* tr_gamma, const float* tr_beta, float* emit,
                               int64_t N, int H, int W, int Cin, int Cout, double* stats_ws, float eps, float momentum, float* mean,
                               float* invstd, float* running_mean, float* running_var, void* stream) {
    return conv_impl<float>(in, w_packed, out, residual, res_mask, tr_mean, tr_invstd, tr_gamma, tr_beta, emit, N, H, W, Cin, Cout, stats_ws,
                            eps, momentum, mean, invstd, running_mean, running_var, stream);
}

// Inference form (the layer1 convs of the patch-scoring pass, models.py:224 in eval mode): out = act(conv(in) + shift (+ residual)) with the
// folded BatchNorm's SCALE folded into the packed filter (ssad_conv3x3_fw_pack_scaled) and its shift added here; 16 x 16 maps of 64
// channels run two maps per tile.  out_hwnc: the output is written position-major [H][W][N][C] (in and residual are NHWC).
extern "C" int ssad_conv3x3_fw_eval_ok(int64_t N, int H, int W, int Cin, int Cout) {
    if (!shape_ok(N, H, W, Cin, Cout)) return 0;
    const GeoW g = geometry_w(N, H, W, Cout);
    return g.ntiles * g.gy >= 512 || Cin >= 256;
}
extern "C" int ssad_conv3x3_fw_pack_scaled(const float* w_ohwi, const float* scale, float* dst, int Cout, int Cin, void* stream) {
    SSAD_CHECK_ARG(w_ohwi && dst && Cout > 0 && Cin > 0 && Cout % 64 == 0 && Cin % 64 == 0, "bad argument (channel counts multiples of 64)");
    PackTable t;
    t.n = 1;
    t.e[0][0] = 0; t.e[0][1] = 0; t.e[0][2] = Cout; t.e[0][3] = Cin; t.e[0][4] = 0; t.e[0][5] = 0;
    const int64_t blocks = cdiv64((int64_t)Cout * 9 * Cin / 4, 256);
    hipLaunchKernelGGL(pack_hw_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w_ohwi, dst, t, scale);
    SSAD_CHECK_LAUNCH();
    return 0;
}
extern "C" int ssad_conv3x3_fw_eval(const float* in, const float* w_packed, float* out, const float* shift, const float* residual, int relu,
                                    int64_t N, int H, int W, int Cin, int Cout, int out_hwnc, void* stream) {
    SSAD_CHECK_ARG(shift, "null shift");
    return conv_impl<float>(in, w_packed, out, residual, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, nullptr, 0.f,
                            0.f, nullptr, nullptr, nullptr, nullptr, stream, shift, relu, out_hwnc);
}
