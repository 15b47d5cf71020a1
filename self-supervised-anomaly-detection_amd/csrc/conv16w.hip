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
//   * one barrier per (tile, chunk) = per 36 steps;
//   * (round 6) halo pieces and filter fragments are BUFFER loads -- scalar base, per-lane 32-bit offset, out-of-range offsets read zeros:
//     a VALU instruction costs the matrix pipe ~15-25 cycles whichever wave issues it, and the stagers only advance in the matrix waves'
//     stalls (time stamps: tools/micro/conv32w_trace.hip, -DCONV16W_TRACE).
// Workgroups are persistent over tiles of ONE channel slab (blockIdx.y), so the BatchNorm statistics of the stored output stay in
// registers (per tile in fp32, across tiles in double) and leave as one partial row per workgroup.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#ifndef CONV16W_ABL      // timing ablations (tools/micro): 1 = no weight loads, 2 = no halo loads, 4 = no epilogue, 8 = no MFMAs,
                         // 16 = no activation fragment reads, 64 = the epilogue without its stores
#define CONV16W_ABL 0
#endif

#ifdef CONV16W_TRACE
static long long* conv16w_trace_buf = nullptr;
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
#ifdef CONV16W_TRACE        // tools/micro: 10 ns time stamps of workgroup CONV16W_TRACE's eight waves, 1 024 (tag, time) pairs each
    long long* trace;
#endif
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The barrier between fills orders LDS traffic only (the halo stages): __syncthreads() also drains every wave's vector-memory queue -- the
// stagers' halo loads of two fills ahead and the matrix waves' filter fragments in flight -- at each of the 16-32 barriers.  (Measured
// neutral on both steps, round 6: the waits it removes were hidden behind the partner waves.)
__device__ __forceinline__ void fill_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

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
    // filter fragments are requested DB steps ahead (an fp32 step is 32 MFMAs of 64 cycles).  Float, 64-channel form: six -- its 228
    // registers leave room, and with three the matrix waves waited ~110 cycles per step for fragments (614 -> 602 us on zeros, round 6;
    // four steps on the 128-channel form, 241 + 8 registers: nothing)
    constexpr int DB = F32 ? (WN == 1 ? 6 : 3) : 6;
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
#ifdef CONV16W_TRACE
    int tr_i = 0;
    const bool tr_on = (int)blockIdx.x == CONV16W_TRACE && blockIdx.y == 0 && lane == 0 && p.trace;
    long long* const tr_p = p.trace + wave * 2048;         // 1 024 (tag, time) pairs per wave
#define TR(tag) do { if (tr_on && tr_i < 1020) { tr_p[2 * tr_i] = (long long)(tag) | (long long)clock64() << 8; tr_p[2 * tr_i + 1] = wall_clock64(); ++tr_i; } } while (0)
#else
#define TR(tag) do { } while (0)
#endif
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
#ifndef CONV16W_PRIO_MODE
#define CONV16W_PRIO_MODE 1
#endif
    if (CONV16W_PRIO_MODE == 1 && __builtin_amdgcn_readfirstlane(tid) < NMW * 64) __builtin_amdgcn_s_setprio(3);
    if (CONV16W_PRIO_MODE == 2 && __builtin_amdgcn_readfirstlane(tid) >= NMW * 64) __builtin_amdgcn_s_setprio(3);

    if (wave >= NMW) {
        // =====================================================================================================================
        // stager waves: fill f -> halo[f & 1], one fill ahead of the matrix waves; the loads of fill f + 2 are requested BEFORE the
        // barrier that ends fill f (they have a whole fill of matrix work to arrive: HBM latency is never on the barrier's path)
        // =====================================================================================================================
        // The stagers share two SIMDs with matrix waves: every instruction here delays a matrix wave, and the slowest wave sets the
        // pace at the barrier (first form: ~1 500 instructions of index arithmetic per fill = 1.3 us per fill on every layer).
        // Everything that depends only on (lane, piece) is computed ONCE: byte offset from the halo's first pixel, LDS byte offset,
        // halo coordinates, interior flag; per fill there remain the in-image tests and the uniform base address.
        // Instruction budget (round 6, time stamps inside the kernel -- tools/micro/conv32w_trace.hip): a VALU instruction of a stager
        // cannot issue while an MFMA of the matrix wave on its SIMD is in the pipe -- every one of them waits for the running MFMA (64
        // cycles in the float form) and then takes the pipe's next slot.  With ~520 VALU instructions per fill (per-piece in-image tests,
        // 64-bit address arithmetic, selects around conditional loads, two 64-bit divisions) the stagers needed a whole fill of matrix
        // work (17-33 us) for 1.7 us of their own, arrived ~1 us AFTER the matrix waves at every barrier, and their instructions showed up
        // one for one in the matrix stream.  So: loads and the emitted activation go through BUFFER instructions (uniform base in SGPRs,
        // one precomputed 32-bit offset per piece; an offset beyond the buffer reads zeros / drops the store: padding and missing images
        // cost no instruction), the per-piece offsets are rebuilt once per TILE, and the tile walk is two scalar cursors (one for the
        // loads, one for the LDS writes) instead of divisions per fill.  A plain fill is now NR loads + NR LDS writes.
        constexpr unsigned OOB = 0x80000000u;               // buffer size: offsets from here on are out of range
        constexpr int SRD3 = 0x00020000;                    // raw buffer, 32-bit data format
#ifndef CONV16W_NT                                          // experiment: 1 = halo loads non-temporal, 2 = output stores, 3 = both
#define CONV16W_NT 0
#endif
        constexpr int LDAUX = (CONV16W_NT & 1) ? 2 : 0;     // aux bit 1 = nt
        const int sl = (wave - NMW) * 64 + lane;
        const int piece = sl % PPR;                         // SL is a multiple of PPR: a lane stages the same piece of every pixel
        u32x4 reg[2][NR];                                   // two sets: a fill's loads are requested TWO fills before they are written
        unsigned goff[NR];                                  // bytes from the halo's pixel (-1, -1) (TW8: from image 4 tile, pixel (0, 0))
        unsigned loff[NR];                                  // bytes from the start of a halo stage
        unsigned goffr[NR];                                 // the same in the residual tensor (Cout channels per pixel)
        unsigned voff_i[NR], voff_r[NR], voff_m[NR];        // per TILE of the load cursor: input / residual / mask-byte offsets (or OOB)
        unsigned voff_e[NR];                                // emitted activation: interior pixels only
        unsigned rmask[2][NR];                              // the residual's pass bits of each piece: 4 bits (float), 2 x 4 in two bytes (half)
        unsigned valid = 0, inner = 0, m_top = 0, m_bot = 0, m_left = 0, m_right = 0, m_i1 = 0, m_i2 = 0, m_i3 = 0;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int hp = (q * SL + sl) / PPR;             // TW8: hp = 64 image + 8 y + x (interior pixels only)
            if (hp < NSP) valid |= 1u << q;
            if (TW8) {
                goff[q] = (unsigned)((hp * p.Cin + piece * E) * sizeof(T));
                goffr[q] = (unsigned)((hp * p.Cout + piece * E) * sizeof(T));
                loff[q] = (unsigned)((((hp >> 6) * 100 + (((hp >> 3) & 7) + 1) * 10 + (hp & 7) + 1) * LDP + piece * E) * sizeof(T));
                inner |= 1u << q;
                if ((hp >> 6) >= 1) m_i1 |= 1u << q;        // pieces of images >= 1 / 2 / 3 of the tile's four
                if ((hp >> 6) >= 2) m_i2 |= 1u << q;
                if ((hp >> 6) >= 3) m_i3 |= 1u << q;
            } else {
                const int img = TWP ? hp / (18 * HW_) : 0, hq = hp - img * 18 * HW_;
                const int hy = hq / HW_, hx = hq - HW_ * hy;
                goff[q] = (unsigned)((((img * p.H + hy) * p.W + hx) * p.Cin + piece * E) * sizeof(T));
                goffr[q] = (unsigned)((((img * p.H + hy) * p.W + hx) * p.Cout + piece * E) * sizeof(T));
                loff[q] = (unsigned)((hp * LDP + piece * E) * sizeof(T));
                if (hy >= 1 && hy <= 16 && hx >= 1 && hx <= TWX) inner |= 1u << q;
                if (hy == 0) m_top |= 1u << q;              // halo rows / columns that fall outside the image when the tile touches
                if (hy == 17) m_bot |= 1u << q;             // that edge (maps are whole numbers of tiles)
                if (hx == 0) m_left |= 1u << q;
                if (hx == TWX + 1) m_right |= 1u << q;
                if (img == 1) m_i1 |= 1u << q;
            }
        }
        inner &= valid;
#pragma unroll
        for (int q = 0; q < NR; ++q) voff_e[q] = ((inner >> q) & 1u) ? goff[q] : OOB;
        const bool lastvalid = (valid >> (NR - 1)) & 1u;    // pieces 0 .. NR - 2 exist in every lane

        // a cursor: fill k of this workgroup's tile number ti, with the tile's geometry
        struct Cur { int ti, k, y0, x0, nimg, pix; };       // pix: pixel index of the halo's (-1, -1) (TW8: of image 4 tile's first pixel);
                                                            // may be negative -- only in-image pixels are dereferenced
        auto enter = [&](Cur& c) {
            const unsigned tile = blockIdx.x + (unsigned)c.ti * gridDim.x;
            if (TW8) {
                c.y0 = c.x0 = 0;
                c.nimg = p.N - (int)(4 * tile);             // images of this tile that exist
                c.pix = (int)(4 * tile) * 64;
            } else if (TWP) {
                c.y0 = c.x0 = 0;
                c.nimg = p.N - (int)(2 * tile);
                c.pix = ((int)(2 * tile) * p.H - 1) * p.W - 1;
            } else {
                const unsigned n0 = tile / (unsigned)tpi, rem = tile - n0 * (unsigned)tpi;
                const unsigned ty = rem / (unsigned)p.tiles_x;
                c.y0 = (int)ty * 16;
                c.x0 = (int)(rem - ty * (unsigned)p.tiles_x) * TWX;
                c.nimg = 1;
                c.pix = ((int)n0 * p.H + c.y0 - 1) * p.W + c.x0 - 1;
            }
        };
        auto ok_mask = [&](const Cur& c) -> unsigned {      // pieces of the tile's halo that lie inside an existing image
            unsigned bad = 0;
            if (TW8) {
                if (c.nimg < 4) bad = c.nimg <= 1 ? m_i1 : c.nimg == 2 ? m_i2 : m_i3;
            } else {
                if (c.y0 == 0) bad |= m_top;
                if (c.y0 + 16 == p.H) bad |= m_bot;
                if (c.x0 == 0) bad |= m_left;
                if (c.x0 + TWX == p.W) bad |= m_right;
                if (TWP && c.nimg < 2) bad |= m_i1;
            }
            return valid & ~bad;
        };
        // order of a tile's fills: input chunk 0, residual chunk 0, input 1, residual 1, ... -- a residual fill is four steps of matrix
        // work, and behind a long input fill its loads (requested two fills ahead) have time to arrive; whatever one kind has more of
        // follows (first form: all residual fills after the last input chunk, +59 us on the fp32 layer2 shape).
        // chunk >= nchunks: a residual fill (chunk - nchunks of the workgroup's channel slab)
        const int mpair = p.residual ? (p.nchunks < RF ? p.nchunks : RF) : 0;
        auto chunk_of = [&](int k) -> int {
            if (k < 2 * mpair) return (k & 1) ? p.nchunks + (k >> 1) : (k >> 1);
            return (p.nchunks > mpair ? 0 : p.nchunks) + (k - mpair);
        };
        Cur L = {0, 0, 0, 0, 0, 0}, Wr = {0, 0, 0, 0, 0, 0};
        unsigned wmask = 0;                                 // in-image pieces of the WRITE cursor's tile (the transform skips the others)
        auto retarget = [&]() {                             // the load cursor entered a tile: offsets of its in-image pieces
            const unsigned ok = ok_mask(L);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const bool in = (ok >> q) & 1u, rs = in && ((inner >> q) & 1u);     // only the centre tap reads a residual fill
                voff_i[q] = in ? goff[q] : OOB;
                voff_r[q] = rs ? goffr[q] : OOB;
                // the identity-branch gradient is (dy, nibble mask), never materialised: one mask byte per channel quad, i.e. per
                // 16-byte piece one byte (float) or two (half: 8 channels), at the piece's byte offset / 16 (/ 8)
                voff_m[q] = rs ? goffr[q] >> (F32 ? 4 : 3) : OOB;
            }
        };
        enter(L); retarget();
        enter(Wr); wmask = ok_mask(Wr);
        auto load_fill = [&](auto set_tag) {
            constexpr int SET = decltype(set_tag)::value;
            const int chunk = chunk_of(L.k);
            const bool res = chunk >= p.nchunks;
            const int C = res ? p.Cout : p.Cin;
            const int c0 = res ? (int)blockIdx.y * (64 * WN) + (chunk - p.nchunks) * CK : chunk * CK;
            const int64_t boff = ((int64_t)L.pix * C + c0) * (int64_t)sizeof(T);          // bytes; a multiple of 16
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(res ? p.residual : p.in) + boff), 0, (int)OOB, SRD3);
            if (CONV16W_ABL & 2) {
#pragma unroll
                for (int q = 0; q < NR; ++q) reg[SET][q] = u32x4{0u, 0u, 0u, 0u};
            } else if (res) {
#pragma unroll
                for (int q = 0; q < NR; ++q) reg[SET][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_r[q], 0, LDAUX);
                if (p.res_mask) {
                    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(p.res_mask + (boff >> (F32 ? 4 : 3))), 0, (int)(OOB >> (F32 ? 4 : 3)), SRD3);
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        if (F32) rmask[SET][q] = __builtin_amdgcn_raw_buffer_load_b8(rm, voff_m[q], 0, 0);
                        else rmask[SET][q] = __builtin_amdgcn_raw_buffer_load_b16(rm, voff_m[q], 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < NR; ++q) reg[SET][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_i[q], 0, LDAUX);
            }
            if (++L.k == fpt) { L.k = 0; ++L.ti; enter(L); retarget(); }
        };
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        auto write_fill = [&](auto set_tag) {               // halo stage = set (fill f goes to stage f & 1 and was loaded into set f & 1)
            constexpr int SET = decltype(set_tag)::value;
            const int chunk = chunk_of(Wr.k);
            char* dst = (char*)halo + SET * HALO_H * (int)sizeof(T);
            if (p.tr_mean && chunk < p.nchunks) {
                // producer's train-mode BatchNorm + ReLU on load: bn_apply_fwd's expression in fp32 (two channels per packed
                // instruction); halves: rounded once.  Zero padding pads the TRANSFORMED activation: out-of-image pieces stay zero
                // (they were read as zeros).  The E channels of this lane's piece: read once per fill.
                const int c = chunk * CK + piece * E;
                f32x2 mu[E / 2], sc[E / 2], ga[E / 2], be[E / 2];
#pragma unroll
                for (int k = 0; k < E / 2; ++k) {
                    mu[k] = *(const f32x2*)(trp + c + 2 * k); sc[k] = *(const f32x2*)(trp + p.Cin + c + 2 * k);
                    ga[k] = *(const f32x2*)(trp + 2 * p.Cin + c + 2 * k); be[k] = *(const f32x2*)(trp + 3 * p.Cin + c + 2 * k);
                }
                // interior pixels of the halo: the activation this layer's weight gradient reads (written by channel slab 0; a buffer
                // of size zero drops every store otherwise)
                const bool emit = p.emit && blockIdx.y == 0;
                const int64_t boff = ((int64_t)Wr.pix * p.Cin + chunk * CK) * (int64_t)sizeof(T);
                const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)((char*)(emit ? p.emit : p.out) + (emit ? boff : 0)), 0, emit ? (int)OOB : 0, SRD3);
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    if (!((wmask >> q) & 1u)) continue;
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
                    __builtin_amdgcn_raw_buffer_store_b128(w, re, voff_e[q], 0, 0);
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
                if (q < NR - 1 || lastvalid) *(u32x4*)(dst + loff[q]) = reg[SET][q];
            if (++Wr.k == fpt) { Wr.k = 0; ++Wr.ti; enter(Wr); wmask = ok_mask(Wr); }
        };
        // fill g is loaded into set g & 1 while fill g - 2 (same set, already written) is being consumed: HBM has two fills of matrix
        // work to answer (one was not enough on the 64-channel layer: 18 steps per fill)
        const std::integral_constant<int, 0> S0;
        const std::integral_constant<int, 1> S1;
        if (nfill > 0) { load_fill(S0); write_fill(S0); }
        if (nfill > 1) load_fill(S1);
        if (nfill > 2) load_fill(S0);
        TR(100);
        fill_barrier();
        for (int64_t f = 0; f < nfill; f += 2) {
            TR(101);
            if (f + 1 < nfill) write_fill(S1);
            TR(102);
            if (f + 3 < nfill) load_fill(S1);
            TR(103);
            fill_barrier();
            if (f + 1 >= nfill) break;
            TR(101);
            if (f + 2 < nfill) write_fill(S0);
            TR(102);
            if (f + 4 < nfill) load_fill(S0);
            TR(103);
            fill_barrier();
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
    // the fragments as BUFFER loads: the wave's first fragment in scalar registers, the lane's 16 bytes a constant offset, the step's
    // fragment a scalar offset -- no address arithmetic in the matrix stream (the pointer form: one 64-bit add per fragment and tap)
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wp + (int64_t)__builtin_amdgcn_readfirstlane(co0 / 32) * 9 * KB * (64 * E)), 0, (int)0x80000000u, 0x00020000);
    const unsigned blane = (unsigned)(lane * 16);
    const int bcol = 9 * KB * (64 * E) * (int)sizeof(T);          // bytes between the two 32-channel fragments of a wave

    // Every workgroup walks the nine taps in its own rotation (conv16.hip: persistent workgroups run in lockstep, with one common order
    // all of them ask the L2 for the same lines at the same moment).  fp32 accumulation order differs between workgroups by the
    // rotation only: deterministic for a given launch geometry.
    // (fp32: no rotation -- a fragment is requested every 2 048 cycles, and one common tap order keeps every output value's summation
    // order independent of the launch geometry)
#ifdef CONV16W_NO_ROT
    const int rot = 0;
#else
    const int rot = F32 ? 0 : (int)((blockIdx.x + blockIdx.y) % 9);
#endif
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
    frag_t bdummy[DB][2];                                  // ablation 32: the fragment loads are issued, nobody waits for them within a fill
    auto load_b = [&](int chunk, int s, int set) {        // s: step within a chunk (compile-time after unrolling)
        if (CONV16W_ABL & 1) return;
        const int off = tapB[s / KS] + (chunk * KS + (s % KS)) * (64 * E);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (CONV16W_ABL & 32) bdummy[set][j] = *(const frag_t*)(bptr[j] + off);
            else breg[set][j] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(brs, blane, off * (int)sizeof(T) + j * bcol, 0));
        }
    };
    if (CONV16W_ABL & 33) {
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
    TR(0);
    fill_barrier();                                       // fill 0 is staged
    TR(1);

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
        if (CONV16W_ABL & 32) {
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(bdummy[d][j]));
        }
        TR(2);
        fill_barrier();                                   // everyone is done with halo[f & 1]; fill f + 1 is staged
        TR(3);
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
            fill_barrier();
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
        TR(4);
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
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        f32x2v ps[2] = {{0.f, 0.f}, {0.f, 0.f}}, pq[2] = {{0.f, 0.f}, {0.f, 0.f}};
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
                // exact-fp32 step: values stored as they are.  Statistics: two packed fp32 accumulators per column and tile (32 values
                // each: v_pk_add_f32 / v_pk_fma_f32, one instruction per value instead of three -- in this kernel every VALU instruction
                // is time the matrix pipe stands still, ~14 cycles each by the time stamps), added in double across tiles; the rounding
                // of a 32-value fp32 partial sum is ~1e-8 of the final mean / variance.  Inference (p.shift): the folded BatchNorm's
                // shift and the ReLU instead, no statistics.
                // (one uniform branch per accumulator row, not per register pair: with `if (p.shift)` inside, hipcc emitted a branch and
                // its register shuffles for each of the 64 pairs)
                auto rows = [&](auto infer_tag) __attribute__((always_inline)) {
                    constexpr bool INFER = decltype(infer_tag)::value;
                    float sh[2] = {0.f, 0.f};
                    if (INFER) { sh[0] = p.shift[co0 + r]; sh[1] = p.shift[co0 + 32 + r]; }
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            f32x2v v = {acc[m][j][e], acc[m][j][e + 1]};
                            if (INFER) {
                                v += f32x2v{sh[j], sh[j]};
                                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
                            } else {
                                ps[j] += v;
                                pq[j] = v * v + pq[j];
                            }
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const int ee = e + k;
                                const int eo = TW8 ? (ee >> 2) * rowst + (ee & 3) * pixst : (ee >> 3) * rowst + ((ee & 3) + 8 * ((ee >> 2) & 1)) * pixst;
                                // uniform base (scalar registers) + this lane's constant 32-bit offset
                                *(float*)(ob + (int64_t)eo * 4 + 128 * j + lo) = v[k];
                            }
                        }
                    }
                };
                if (p.shift) rows(std::true_type());
                else rows(std::false_type());
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
        for (int j = 0; j < 2; ++j) {
            if (F32) { st0[j] += (double)ps[j][0] + (double)ps[j][1]; st1[j] += (double)pq[j][0] + (double)pq[j][1]; }
            else { st0[j] += (double)fs[j]; st1[j] += (double)fq[j]; }
        }
        TR(5);
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
#ifdef CONV16W_TRACE
    p.trace = conv16w_trace_buf;
#endif
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
    static const int min_pairs = getenv("SSAD_CONV32W_MIN") ? atoi(getenv("SSAD_CONV32W_MIN")) : 512;
    return g.ntiles * g.gy >= min_pairs || Cin >= 256;
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

// The exact-fp32 form of the same kernel (fp32 tensors, v_mfma_f32_32x32x2_f32, statistics per tile in packed fp32, across tiles in double): the 3 x 3 / stride 1
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
