// Implicit-GEMM convolution / linear layer on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   out[m][co] = act( (sum_k A[m][k] * Wt[co][k]) * scale[co] + shift[co] + residual[m][co] )
//   m = output pixel, k = (ky, kx, ci) with ci fastest (OHWI weights).
//
// Replaces the nn.Conv2d/BatchNorm2d/ReLU/residual chains of the torchvision BasicBlocks and the
// nn.Linear/BatchNorm1d/ReLU chains of the projection head that the reference runs through cuDNN /
// MKLDNN (src/self_supervised/models.py:224, :247-249); in transposed-gather mode (ssad_conv_igemm_dgrad)
// the input gradient of those layers under loss.backward() (src/self_supervised/tools.py:270, :303).
//
// Design (MI355X): 256-thread workgroups = 4 waves, one per SIMD; each wave owns TM x TN accumulator tiles of 32x32
// (f32x16 each).  A (gathered activations) and B (weights) K-slices of 32 floats are staged global -> registers ->
// LDS with 144-byte rows (128 B + 16 B pad: conflict-free ds_read_b128 for the "row = lane&31" fragment pattern),
// double buffered, one barrier per K-step.  Each lane reads 4 consecutive k per ds_read_b128; lane half h takes
// k = 8*kk + 4*h + e for the e-th MFMA of a chunk, the same permutation for A and B, so the contraction is unchanged.
//
// The K-loop carries no integer division and no branch: every staged row has a precomputed base pointer and a
// bitmask of in-bounds filter taps; a tap's offset is a wave-uniform scalar; out-of-bounds (zero padding / tail)
// rows load from a zero page instead of branching around the load.  The epilogue goes through LDS so that every
// thread stores (and reads the residual as) 16-byte pieces of contiguous output rows, with all residual loads in
// flight at once.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

// Ablation switch for tools/micro/igemm_ablate.hip; always 0 in the library build.
#ifndef IGEMM_ABL
#define IGEMM_ABL 0
#endif
// K-loop schedule of the exact-fp32 path: 0 = loads ahead of the first MFMA, next stage stored after the last MFMA of a K-step
// (rounds 1-2); 3 = loads (chunk 0) AND stores (last chunk) between MFMAs; 4 = the same with two staging register sets: loads
// two K-steps ahead, stores in chunk 1 of the following K-step
#ifndef IGEMM_VAR
#define IGEMM_VAR 4
#endif
// 1 = direct (LDS-free) epilogue where the output rows of a tile are equally spaced; 0 = always the LDS-transpose epilogue
#ifndef IGEMM_EPI
#define IGEMM_EPI 1
#endif


// 1 = the staging loads are BUFFER loads: a uniform base in scalar registers (tap / channel-chunk offsets folded into it per K-step) plus
// one precomputed 32-bit offset per staged row; rows beyond the tensor carry an offset beyond the buffer and read zeros.  In the K loop
// every VALU instruction is matrix-pipe time (the co-resident wave does not fill the gap: ~15-25 cycles each, tools/micro/conv32w_trace):
// the pointer form spent ~3.7 of them per 16-byte piece (select against the zero page, 64-bit add).  0 = the pointer form.
#ifndef IGEMM_BUFLD
#define IGEMM_BUFLD 1
#endif

// Phase time stamps per workgroup for tools/micro/igemm_var.hip (-DIGEMM_TRACE=1); never set in the library build.
#ifndef IGEMM_TRACE
#define IGEMM_TRACE 0
#endif
#if IGEMM_TRACE
__device__ unsigned long long* g_ig_trace;      // [workgroups][8]
#define IG_STAMP(i) do { if (threadIdx.x == 0 && g_ig_trace) g_ig_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define IG_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int KALIGN = 32;   // Cin granularity every instantiation accepts

__device__ __attribute__((aligned(16))) float g_zero_page[8] = {0, 0, 0, 0, 0, 0, 0, 0};

struct ConvParams {
    const float* in;
    const float* wt;
    float* out;
    const float* scale;
    const float* shift;
    const float* residual;
    const uint8_t* res_mask;   // optional nibble mask on the residual (one byte per channel quad of an output row)
    int64_t M;          // N*Ho*Wo
    int64_t N;          // samples
    int H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, relu;
    int K;              // KH*KW*Cin
    int hwnc;           // activations (in, out, residual) laid out [H][W][N][C] instead of [N][H][W][C]
    int cls_start[5];   // TS == 2: first workgroup of the c-th output-parity class, (py, px) = (cls_id[c] >> 1, cls_id[c] & 1)
    int cls_id[4];
    double* stats;      // optional [gridDim.x][2][Cout]: per-workgroup column sums / sums of squares of the raw output
    // POS, pos_lpt != 0: workgroups are numbered position-major with the positions sorted by their in-bounds tap count,
    // heaviest first (pos_tab), sample groups fastest -- longest-processing-time-first for the in-order dispatcher
    int skip_lo, skip_hi;       // POS: output positions with skip_lo <= oy, ox <= skip_hi are NOT computed (their workgroups retire at once):
                                // the caller fills them from elsewhere (ssad_patch_gather_hwnc).  skip_lo > skip_hi: none.
    int pos_lpt;
    int pos_sg;         // sample groups per position
    int pos_chunk;      // sample groups per chunk (workgroups along x = ceil(pos_sg / pos_chunk) * pos_chunk * Ho * Wo)
    unsigned char pos_tab[64];
};

// TS   : 1 = convolution; 2 = transposed gather (dgrad of a stride-2 conv): tap (ky,kx) reads in[(oy-pad+ky)/2]
//        only where the numerator is a non-negative even number.
// POS  : a workgroup's rows are BM different samples at ONE output position, so the set of in-bounds taps is
//        workgroup-uniform and taps in the zero padding are skipped as whole K-steps (exact: only x*0 is dropped).
// BK   : floats per K-step (32: 144-byte LDS rows; 16: 80-byte rows -- both conflict-free for ds_read_b128).
// DB   : double-buffered LDS stages (one barrier per K-step) or a single stage (two barriers, half the LDS:
//        more workgroups per CU).
// BF   : 1 = operands are rounded to bf16 while staging (fp32 in HBM, fp32 accumulate): v_mfma_f32_32x32x16_bf16 runs
//        16x the fp32 rate, the kernel becomes load-bound.  Trainer(precision="bf16").
//        2 = the same with fp16 operands (v_mfma_f32_32x32x16_f16, same rate): what the reference's fp16 autocast
//        (pl.Trainer(precision=16), tools.py:263) feeds its convolutions -- 11-bit significands, fp32 accumulation.
//        3 = split-bf16 emulation of the fp32 product: every operand x is staged as hi = bf16(x), lo = bf16(x - hi)
//        (16 significant bits together; a row keeps its fp32 footprint: [BK hi | BK lo] bf16) and each 32x32x16 tile
//        takes three MFMAs, lo*hi + hi*lo + hi*hi, accumulated in fp32: per-product relative error ~2^-17 instead of
//        2^-24 at 3/16 of the fp32-MFMA time ("bf16x3"; opt-in, parity-tested at the same 1e-4 bar).
//        6 = three-way split hi + mid + lo (24 significant bits, rows [BK hi | BK mid | BK lo]) and the six products
//        of weight >= 2^-18 (hh, hm, mh, hl, lh, mm): what is dropped is ~2^-25 of the product, i.e. fp32-faithful
//        products at 6/16 of the fp32-MFMA time ("bf16x6").
// IO   : true = the activation tensors (in, out, residual) AND the weights are stored as halves (fp16 operands only, BF == 2): the
//        precision-16 training step with its tensors as torch.autocast keeps them.  A 16-byte piece is then 8 consecutive k, staged
//        without conversion; outputs are rounded once, in the epilogue; BatchNorm statistics are those of the stored halves.
template <int BM, int BN, int TM, int TN, int BK, int TS, bool POS, bool DB = true, int BF = 0, bool IO = false>
__global__ __launch_bounds__((BM / (32 * TM)) * (BN / (32 * TN)) * 64, (TM * TN > 8 || (TM * TN == 8 && BK >= 32)) ? 1 : 2)
void conv_igemm_f32_kernel(ConvParams p) {
    static_assert(!IO || BF == 2, "half tensors go with fp16 operands");
    using io_t = std::conditional_t<IO, hf, float>;
    constexpr int PE = IO ? 8 : 4;  // elements per 16-byte piece
    constexpr int NT = (BM / (32 * TM)) * (BN / (32 * TN)) * 64;   // threads: one wave per (32 TM) x (32 TN) sub-tile
    // LDS row stride in elements (f32: 144 / 80 bytes; bf16: 80 / 48 bytes; bf16x3: hi and lo halves, the f32 bytes)
    constexpr int LDK = BF == 6 ? 3 * BK + 8 : (BF == 3 ? 2 * BK + 8 : (BF ? BK + 8 : BK + 4));
    constexpr int ESZ = BF ? 2 : 4;
    constexpr int CPR = BK / PE;    // 16-byte chunks per staged row
    constexpr int RPP = NT / CPR;   // rows staged per pass
    constexpr int WN = BN / (32 * TN);
    constexpr int AR = BM / RPP;    // 16-byte chunks of A staged per thread per K-step
    constexpr int BR = BN / RPP;
    static_assert(AR >= 1 && BR >= 1, "tile too small for the staging pattern");
    constexpr int STAGE = (BM + BN) * LDK * ESZ / 4;     // floats
    constexpr int LDC = BN + 4;     // epilogue tile row stride (floats)
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * BN;
    IG_STAMP(0);
#if IGEMM_TRACE
    if (threadIdx.x == 0 && g_ig_trace)
        g_ig_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] =
            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    const int sc = tid % CPR, sr = tid / CPR;
    const int HoWo = p.Ho * p.Wo;
    const int ntaps = p.KH * p.KW;
    const int64_t in_sp = p.hwnc ? p.N * p.Cin : (int64_t)p.Cin;                       // floats between pixels
    const int64_t in_sn = p.hwnc ? (int64_t)p.Cin : (int64_t)p.H * p.W * p.Cin;        // floats between samples
    const io_t* zero = (const io_t*)g_zero_page;
    const io_t* const tin = (const io_t*)p.in;
    const io_t* const twt = (const io_t*)p.wt;

    // ---- row geometry ----
    int64_t m0;                 // first flattened row (normal) / first sample (POS)
    int pos = 0;
    unsigned tapmask = ntaps >= 32 ? ~0u : ((1u << ntaps) - 1);     // taps this workgroup iterates over
    if (POS) {
        // Engine-aware mapping.  Workgroups are dealt round-robin to the 8 XCDs and, inside an XCD, to its 4 shader
        // engines: stream = block % 32 is served by one (XCD, SE) pair for the whole launch (measured: with
        // pos = block % HoWo the heavy interior positions pin to the same engines and tap skipping buys nothing).
        // Positions differ in work (corner 4 taps .. interior 9), so every stream sweeps ALL positions of its own
        // sample groups (nb = 32*q + stream): equal work per engine, and a sample group's maps stay in one XCD's L2.
        if (p.pos_lpt == 2) {
            // XCD-local sweep (round 4).  Workgroups are dealt round-robin to the 8 XCDs, so x = block % 8 names the XCD (which
            // one is not fixed, only that blocks b and b + 8 share it) and j = block / 8 counts the workgroups of that XCD in
            // dispatch order.  An XCD takes pos_chunk sample groups at a time through ALL positions, heaviest position first:
            // the ~64 workgroups resident on its 32 CUs are then the positions of the same few sample groups, whose input maps
            // (4 MB at 8 x 8 x 128 channels) fit the XCD's 4 MB L2 while all nine taps of every position read them -- instead
            // of one fetch from the fabric per filter row (profiles/r03_traffic.json: 1.52 x the algorithmic bytes).
            const unsigned x = blockIdx.x & 7, j = blockIdx.x >> 3;
            const unsigned per_blk = (unsigned)p.pos_chunk * (unsigned)HoWo;
            const unsigned blk = j / per_blk, r = j - blk * per_blk;
            const unsigned pi = r / (unsigned)p.pos_chunk, g = r - pi * (unsigned)p.pos_chunk;
            const unsigned sg = (blk * (unsigned)p.pos_chunk + g) * 8u + x;
            if (sg >= (unsigned)p.pos_sg) return;
            pos = p.pos_tab[pi];
            m0 = (int64_t)sg * BM;
        } else if (p.pos_lpt) {
            // Positions differ in work (corner 4 taps .. interior 9) and workgroups are handed out in order: all the
            // 9-tap positions first, the 4-tap corners last, every position over all sample groups (consecutive
            // workgroups -> all XCDs / engines see the same mix), so the launch ends on its shortest workgroups.
            // ... chunk by chunk of pos_chunk sample groups, so that the taps of neighbouring positions find a chunk's
            // input maps in the Infinity Cache / L2 instead of fetching them from HBM once per tap
            const unsigned per_chunk = (unsigned)p.pos_chunk * (unsigned)HoWo;
            const unsigned chunk = blockIdx.x / per_chunk, in_chunk = blockIdx.x - chunk * per_chunk;
            const unsigned first = chunk * (unsigned)p.pos_chunk;
            const unsigned here = min((unsigned)p.pos_chunk, (unsigned)p.pos_sg - first);      // sample groups in this chunk
            // (the last chunk is shorter: its workgroups are numbered over `here` groups, the surplus ones retire)
            const unsigned pi = in_chunk / here;
            if (pi >= (unsigned)HoWo) return;
            pos = p.pos_tab[pi];
            m0 = (int64_t)(first + in_chunk - pi * here) * BM;
        } else {
        const int stream = blockIdx.x & 31;
        const int64_t j = blockIdx.x >> 5;
        const int64_t q = j / HoWo;
        pos = (int)(j - q * HoWo);
        m0 = (q * 32 + stream) * BM;
        }
        if (m0 >= p.N) return;
        const int oy = pos / p.Wo, ox = pos - oy * p.Wo;
        if (oy >= p.skip_lo && oy <= p.skip_hi && ox >= p.skip_lo && ox <= p.skip_hi) return;
        tapmask = 0;
        for (int ky = 0, t = 0; ky < p.KH; ++ky)
            for (int kx = 0; kx < p.KW; ++kx, ++t) {
                const int y = oy * p.stride - p.pad + ky, x = ox * p.stride - p.pad + kx;
                if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) tapmask |= 1u << t;
            }
    } else {
        m0 = (int64_t)blockIdx.x * BM;
    }
    // TS == 2 (dgrad of a stride-2 conv): rows are grouped by the parity class of the output pixel.  Within a class
    // the taps whose numerator (oy - pad + ky) is even are the same for every row, so the other 3/4 (3x3) or 3/4 (1x1)
    // of the taps -- which would only multiply structural zeros -- are skipped as whole K-steps.
    int cpy = 0, cpx = 0, Hc = 1, Wc = 1;
    int64_t Mc = 0;
    if (TS > 1) {
        int c = 0;
        while (c < 3 && (int)blockIdx.x >= p.cls_start[c + 1]) ++c;
        cpy = p.cls_id[c] >> 1;      // classes come heaviest first (3 x 3 / pad 1: (1, 1) keeps 4 of 9 taps, (0, 0) one)
        cpx = p.cls_id[c] & 1;
        Hc = (p.Ho - cpy + 1) / 2;
        Wc = (p.Wo - cpx + 1) / 2;
        Mc = p.N * Hc * Wc;
        m0 = (int64_t)((int)blockIdx.x - p.cls_start[c]) * BM;
        tapmask = 0;
        for (int ky = 0, t = 0; ky < p.KH; ++ky)
            for (int kx = 0; kx < p.KW; ++kx, ++t)
                if (((cpy - p.pad + ky) & 1) == 0 && ((cpx - p.pad + kx) & 1) == 0) tapmask |= 1u << t;
    }
    // flattened row -> (valid, sample, oy, ox) of the OUTPUT pixel it produces
    auto decode_row = [&](int64_t m, int64_t& n, int& oy, int& ox) -> bool {
        if (TS > 1) {
            const bool ok = m < Mc;
            const int64_t mm = ok ? m : 0;
            n = mm / (Hc * Wc);
            const int rem = (int)(mm - n * (Hc * Wc));
            const int a = rem / Wc;
            oy = 2 * a + cpy;
            ox = 2 * (rem - a * Wc) + cpx;
            return ok;
        }
        const bool ok = m < p.M;
        const int64_t mm = ok ? m : 0;
        n = mm / HoWo;
        const int rem = (int)(mm - n * HoWo);
        oy = rem / p.Wo;
        ox = rem - oy * p.Wo;
        return ok;
    };

    // ---- per-thread staging rows (fixed across the K loop): base pointer at tap (0,0) + in-bounds tap mask ----
    // BUFA: a row's piece is a_org (uniform) + a_voff (this thread's row and 16-byte column, or out of range for rows that do not exist).
    // Position-major: a_org = the workgroup's first sample at the position's tap (0, 0), and the taps a workgroup walks are in bounds
    // for every row.  Pixel-major (NHWC): a_org = the first row's image, one padding row and column before its first pixel, so that
    // every row's offset is non-negative; a row's tap in the padding selects the out-of-range offset instead (one bit test per
    // piece).  Transposed gather (TS == 2): inside a parity class a walked tap reads input pixel (a + dky, b + dkx) for the output pixel
    // (2 a + cpy, 2 b + cpx), with (dky, dkx) the same for every row -- the same uniform-base / per-row-offset split, rows counted in
    // (a, b).  The entry points check that the rows of a tile span less than 2 GB.  BUFB: the same for the weight rows.
    constexpr bool BUFA = IGEMM_BUFLD != 0, BUFB = IGEMM_BUFLD != 0;
    constexpr unsigned OOB = 0x80000000u;   // size given to the buffers: offsets from here on read zeros
    constexpr int SRD3 = 0x00020000;        // raw buffer, 32-bit data format
    int64_t a_org = 0;                      // elements from p.in
    unsigned a_voff[AR], b_voff[BR];        // bytes
    const io_t* a_ptr[AR];
    unsigned a_mask[AR];
    int a_iy[AR], a_ix[AR];                 // only live in the TS == 2 instantiation
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int64_t m = m0 + sr + RPP * i;
        int64_t n;
        int oy, ox;
        bool ok;
        if (POS) {
            ok = m < p.N;
            n = ok ? m : 0;
            oy = pos / p.Wo;
            ox = pos - oy * p.Wo;
        } else {
            ok = decode_row(m, n, oy, ox);
        }
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        a_iy[i] = iy0;
        a_ix[i] = ix0;
        unsigned mk = 0;
        if (POS) {
            mk = ok ? tapmask : 0u;          // every row of the workgroup sits at the same position
        } else if (ok) {
            for (int ky = 0, t = 0; ky < p.KH; ++ky)
                for (int kx = 0; kx < p.KW; ++kx, ++t) {
                    int y = iy0 + ky, x = ix0 + kx;
                    bool v = true;
                    if (TS > 1) {
                        v = y >= 0 && x >= 0 && (y % TS) == 0 && (x % TS) == 0;
                        y /= TS;
                        x /= TS;
                    }
                    if (v && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) mk |= 1u << t;
                }
            mk &= tapmask;
        }
        a_mask[i] = (IGEMM_ABL & 2) ? 0u : mk;      // ablation 2: every activation piece comes from the zero page (no HBM latency)
        a_ptr[i] = TS > 1 ? tin + n * in_sn + sc * PE : tin + n * in_sn + ((int64_t)iy0 * p.W + ix0) * in_sp + sc * PE;
        if (BUFA) {
            if (POS) {
                a_org = m0 * in_sn + ((int64_t)iy0 * p.W + ix0) * in_sp;
                a_voff[i] = (ok && !(IGEMM_ABL & 2)) ? (unsigned)(((int64_t)(sr + RPP * i) * in_sn + sc * PE) * (int64_t)sizeof(io_t)) : OOB;
            } else if (TS > 1) {
                const int64_t nf = (m0 < Mc ? m0 : 0) / (Hc * Wc);                  // image of the workgroup's first row (uniform)
                const int mg = (p.pad + 1) / 2;                                     // dky, dkx >= -mg
                a_org = nf * in_sn - ((int64_t)mg * p.W + mg) * in_sp;
                const int64_t R = (n - nf) * in_sn + ((int64_t)((oy - cpy) / 2 + mg) * p.W + (ox - cpx) / 2 + mg) * in_sp + sc * PE;
                a_voff[i] = ok ? (unsigned)(R * (int64_t)sizeof(io_t)) : OOB;
            } else {
                const int64_t nf = (m0 < p.M ? m0 : 0) / HoWo;                      // image of the workgroup's first row (uniform)
                a_org = nf * in_sn - ((int64_t)p.pad * p.W + p.pad) * in_sp;
                const int64_t R = (n - nf) * in_sn + ((int64_t)(iy0 + p.pad) * p.W + ix0 + p.pad) * in_sp + sc * PE;
                a_voff[i] = ok ? (unsigned)(R * (int64_t)sizeof(io_t)) : OOB;
            }
        }
    }
    const io_t* b_ptr[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int co = n0 + sr + RPP * i;
        b_ptr[i] = co < p.Cout ? twt + (int64_t)co * p.K + sc * PE : nullptr;
        b_voff[i] = co < p.Cout ? (unsigned)(((int64_t)(sr + RPP * i) * p.K + sc * PE) * (int64_t)sizeof(io_t)) : OOB;
    }
    const io_t* const b_org = twt + (int64_t)n0 * p.K;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int cpt = p.Cin / BK;               // K-steps per filter tap
    const int nk = __builtin_popcount(tapmask) * cpt;
    f32x4 ra2[2][AR], rb2[2][BR];          // two staging sets: the exact-fp32 pipeline prefetches two K-steps ahead
    f32x4 (&ra)[AR] = ra2[0];
    f32x4 (&rb)[BR] = rb2[0];

    // state of the next K-step to load: current tap (lowest set bit of ld_mask) and channel chunk -- all scalar
    unsigned ld_mask = tapmask;
    int ld_tap = tapmask ? __builtin_ctz(tapmask) : 0;
    int ld_ky = ld_tap / p.KW, ld_kx = ld_tap - ld_ky * p.KW, ld_cc = 0;
    int64_t ld_koff = 0;
    int ld_woff = 0, ld_ptap = 0, ld_pky = 0, ld_pkx = 0;
    auto load_begin = [&]() {              // scalar offsets of the K-step about to be loaded, then the tap / chunk state moves on
        if (BUFA && TS > 1)     // (numerators of walked taps are even; the shift floors the others, whose pieces are masked off anyway)
            ld_koff = ((int64_t)((cpy - p.pad + ld_ky) >> 1) * p.W + ((cpx - p.pad + ld_kx) >> 1)) * in_sp + ld_cc * BK;
        else
            ld_koff = TS > 1 ? (int64_t)ld_cc * BK : ((int64_t)ld_ky * p.W + ld_kx) * in_sp + ld_cc * BK;
        ld_woff = (ld_tap * cpt + ld_cc) * BK;
        ld_ptap = ld_tap; ld_pky = ld_ky; ld_pkx = ld_kx;
        if (++ld_cc == cpt) {
            ld_cc = 0;
            ld_mask &= ld_mask - 1;
            // past the last K-step the loader wraps (its pieces land in a stage nobody reads): with buffer loads to the first tap the
            // workgroup WALKS -- tap 0 may lie in the padding, and nothing but the per-row offset guards a buffer load
            if (BUFA && POS && !ld_mask) ld_mask = tapmask;
            ld_tap = ld_mask ? __builtin_ctz(ld_mask) : 0;
            ld_ky = ld_tap / p.KW;
            ld_kx = ld_tap - ld_ky * p.KW;
        }
    };
    auto load_piece = [&](int q, int set = 0) {         // one 16-byte piece: q < AR activation rows, then the BR weight rows
#if IGEMM_ABL & 64      // timing experiment only (results are garbage): global -> LDS directly, no VGPR staging, no ds_write
        {
            const io_t* src;
            if (q < AR) {
                const bool ok = (a_mask[q] >> ld_ptap) & 1u;
                src = ok ? a_ptr[q] + ld_koff : zero;
            } else {
                src = b_ptr[q - AR] ? b_ptr[q - AR] + ld_woff : zero;
            }
            float* dst = lds + 2 * STAGE + ((tid >> 6) * (AR + BR) + q) * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            return;
        }
#endif
        if (q < AR) {
            if constexpr (BUFA) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(tin + a_org + ld_koff), 0, (int)OOB, SRD3);
                const unsigned vo = POS ? a_voff[q] : (((a_mask[q] >> ld_ptap) & 1u) ? a_voff[q] : OOB);
                ra2[set][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
            } else {
                const bool ok = (a_mask[q] >> ld_ptap) & 1u;
                const io_t* src = a_ptr[q] + ld_koff;
                if (TS > 1) src += ((int64_t)((a_iy[q] + ld_pky) / TS) * p.W + (a_ix[q] + ld_pkx) / TS) * in_sp;
                ra2[set][q] = *(const f32x4*)(const void*)(ok ? src : zero);
            }
        } else {
            if constexpr (BUFB) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(b_org + ld_woff), 0, (int)OOB, SRD3);
                rb2[set][q - AR] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, b_voff[q - AR], 0, 0));
            } else {
                rb2[set][q - AR] = *(const f32x4*)(const void*)(b_ptr[q - AR] ? b_ptr[q - AR] + ld_woff : zero);
            }
        }
    };
    auto load_step = [&]() {
        load_begin();
#pragma unroll
        for (int q = 0; q < AR + BR; ++q) load_piece(q);
    };
    auto store_step = [&](float* buf) {
        if (BF == 6) {
            __bf16* As = (__bf16*)buf;
            __bf16* Bs = As + BM * LDK;
            auto split3 = [&](const f32x4& x, __bf16* row) {
                bf16x4 hi, mi, lo;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    hi[k] = (__bf16)x[k];
                    const float r1 = x[k] - (float)hi[k];
                    mi[k] = (__bf16)r1;
                    lo[k] = (__bf16)(r1 - (float)mi[k]);
                }
                *(bf16x4*)(row + sc * 4) = hi;
                *(bf16x4*)(row + BK + sc * 4) = mi;
                *(bf16x4*)(row + 2 * BK + sc * 4) = lo;
            };
#pragma unroll
            for (int i = 0; i < AR; ++i) split3(ra[i], As + (sr + RPP * i) * LDK);
#pragma unroll
            for (int i = 0; i < BR; ++i) split3(rb[i], Bs + (sr + RPP * i) * LDK);
            return;
        }
        if (BF == 3) {
            __bf16* As = (__bf16*)buf;
            __bf16* Bs = As + BM * LDK;
            auto split = [&](const f32x4& x, __bf16* row) {
                bf16x4 hi = {(__bf16)x[0], (__bf16)x[1], (__bf16)x[2], (__bf16)x[3]};
                bf16x4 lo = {(__bf16)(x[0] - (float)hi[0]), (__bf16)(x[1] - (float)hi[1]), (__bf16)(x[2] - (float)hi[2]),
                             (__bf16)(x[3] - (float)hi[3])};
                *(bf16x4*)(row + sc * 4) = hi;
                *(bf16x4*)(row + BK + sc * 4) = lo;
            };
#pragma unroll
            for (int i = 0; i < AR; ++i) split(ra[i], As + (sr + RPP * i) * LDK);
#pragma unroll
            for (int i = 0; i < BR; ++i) split(rb[i], Bs + (sr + RPP * i) * LDK);
            return;
        }
        if (BF == 2 && IO) {                 // the pieces are halves already: eight k per 16-byte store
            _Float16* As = (_Float16*)buf;
            _Float16* Bs = As + BM * LDK;
#pragma unroll
            for (int i = 0; i < AR; ++i) *(f32x4*)(void*)(As + (sr + RPP * i) * LDK + sc * 8) = ra[i];
#pragma unroll
            for (int i = 0; i < BR; ++i) *(f32x4*)(void*)(Bs + (sr + RPP * i) * LDK + sc * 8) = rb[i];
            return;
        }
        if (BF == 2) {
            _Float16* As = (_Float16*)buf;
            _Float16* Bs = As + BM * LDK;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                f16x4 v = {(_Float16)ra[i][0], (_Float16)ra[i][1], (_Float16)ra[i][2], (_Float16)ra[i][3]};
                *(f16x4*)(As + (sr + RPP * i) * LDK + sc * 4) = v;
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                f16x4 v = {(_Float16)rb[i][0], (_Float16)rb[i][1], (_Float16)rb[i][2], (_Float16)rb[i][3]};
                *(f16x4*)(Bs + (sr + RPP * i) * LDK + sc * 4) = v;
            }
            return;
        }
        if (BF) {
            __bf16* As = (__bf16*)buf;
            __bf16* Bs = As + BM * LDK;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                bf16x4 v = {(__bf16)ra[i][0], (__bf16)ra[i][1], (__bf16)ra[i][2], (__bf16)ra[i][3]};
                *(bf16x4*)(As + (sr + RPP * i) * LDK + sc * 4) = v;
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                bf16x4 v = {(__bf16)rb[i][0], (__bf16)rb[i][1], (__bf16)rb[i][2], (__bf16)rb[i][3]};
                *(bf16x4*)(Bs + (sr + RPP * i) * LDK + sc * 4) = v;
            }
            return;
        }
        float* As = buf;
        float* Bs = buf + BM * LDK;
#pragma unroll
        for (int i = 0; i < AR; ++i) *(f32x4*)(As + (sr + RPP * i) * LDK + sc * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i) *(f32x4*)(Bs + (sr + RPP * i) * LDK + sc * 4) = rb[i];
    };

    // Exact-fp32, double-buffered instantiations: ONE basic block per K-step with everything that is not an MFMA placed between
    // the wave's own MFMAs.  Measured (profiles/r03_igemm_phases.md): a workgroup ALONE on its CU and two workgroups sharing it
    // take the same time per tile -- the co-resident wave does not fill the gaps of this wave's matrix stream -- so every cycle a
    // wave spends outside MFMA issue is lost: ~60 VALU + 8 loads ahead of the first MFMA and 8 LDS stores behind the last one
    // (the round-2 schedule), and the HBM / L2 latency of a stage that was requested only one K-step (4 096 cycles) before it
    // is needed (~300 cycles per K-step in s_waitcnt vmcnt).  Per K-step of NCH fragment chunks x 4 TM TN MFMAs:
    //   chunk 0    : global loads of the stage TWO K-steps ahead into register set (ks & 1), one 16-byte piece (address
    //                arithmetic included) per MFMA group
    //   chunk 1    : the other register set (requested a whole K-step ago) stored to the other LDS stage, one piece per
    //                MFMA group -- complete long before the barrier that publishes it
    //   all chunks : fragment reads one chunk ahead
    // Loads and stores are unconditional: past the last K-step the loader has wrapped to the first tap (valid addresses or
    // the zero page) and the pieces land in the stage nobody reads any more.
    constexpr bool PIPE = IGEMM_VAR >= 3 && BF == 0 && DB && BK >= 16;
    // two register sets (IGEMM_VAR 3: one set, loads chunk 0, stores last chunk); the 256 x 256 tile (16 accumulator tiles = 256
    // registers per wave, in the AGPR half of the file) has no room for a second staging set
    constexpr bool PIPE2 = PIPE && IGEMM_VAR >= 4 && TM * TN <= 8;
    if (nk > 0) {
        load_step();
        store_step(lds);
        if (PIPE2) {                        // stage 1 goes into flight before the loop: set 1
            load_begin();
#pragma unroll
            for (int q = 0; q < AR + BR; ++q) load_piece(q, 1);
        }
    }
    __syncthreads();
    IG_STAMP(1);

    auto kstep = [&](int ks, auto set_c) {
        constexpr int SET = decltype(set_c)::value;                              // register set the loads of this K-step fill
        constexpr int NCH = BK / 8, NPIECE = AR + BR, NM = 4 * TM * TN;          // chunks, pieces, MFMAs per chunk
        constexpr int TOT = NCH * NM;                                            // MFMAs per K-step
        // one piece per MFMA pair (a 16-byte-per-lane global load costs the matrix stream ~90 cycles, an MFMA is 64): the loads
        // follow MFMAs 1, 3, 5, ... of the K-step, the stores of the other register set start at the next chunk boundary
        constexpr int GAP = TOT >= 4 * NPIECE ? 2 : 1;
        constexpr int S0 = PIPE2 ? (((GAP * NPIECE + NM - 1) / NM) * NM + GAP * NPIECE <= TOT ? ((GAP * NPIECE + NM - 1) / NM) * NM
                                                                                             : TOT - GAP * NPIECE)
                                 : TOT - GAP * NPIECE;                           // one register set: stores as late as possible
        constexpr int SSET = PIPE2 ? SET ^ 1 : 0;                                // register set that is stored
        const float* cur = lds + (ks & 1) * STAGE;
        const float* As = cur + (wm * 32 * TM + r) * LDK + h * 4;
        const float* Bs = cur + BM * LDK + (wn * 32 * TN + r) * LDK + h * 4;
        float* nAs = lds + ((ks + 1) & 1) * STAGE;
        float* nBs = nAs + BM * LDK;
        f32x4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = *(const f32x4*)(As + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = *(const f32x4*)(Bs + j * 32 * LDK);
        load_begin();
#pragma unroll
        for (int kk = 0; kk < NCH; ++kk) {
            const int cu = kk & 1, nx = cu ^ 1;
            if (kk + 1 < NCH && !(IGEMM_ABL & 16)) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nx][i] = *(const f32x4*)(As + i * 32 * LDK + (kk + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nx][j] = *(const f32x4*)(Bs + j * 32 * LDK + (kk + 1) * 8);
            } else if (kk + 1 < NCH) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nx][i] = a[cu][i];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nx][j] = b[cu][j];
            }
            __builtin_amdgcn_sched_barrier(0);
            int m = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = mfma32(a[cu][i][e], b[cu][j][e], acc[i][j]);
                        const int mi = kk * NM + m;                              // index of this MFMA inside the K-step
                        ++m;
                        if (mi % GAP == GAP - 1 && mi / GAP < NPIECE && !(IGEMM_ABL & 4)) {
                            __builtin_amdgcn_sched_barrier(0);
                            load_piece(mi / GAP, PIPE2 ? SET : 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (mi >= S0 && (mi - S0) % GAP == GAP - 1 && (mi - S0) / GAP < NPIECE && !(IGEMM_ABL & 8)) {
                            const int q2 = (mi - S0) / GAP;
                            __builtin_amdgcn_sched_barrier(0);
                            if (q2 < AR) *(f32x4*)(nAs + (sr + RPP * q2) * LDK + sc * 4) = ra2[SSET][q2];
                            else *(f32x4*)(nBs + (sr + RPP * (q2 - AR)) * LDK + sc * 4) = rb2[SSET][q2 - AR];
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(IGEMM_ABL & 32)) __syncthreads();
    };
    if (PIPE) {
        int ks = 0;
        for (; ks + 1 < nk; ks += 2) {
            kstep(ks, std::integral_constant<int, 0>{});
            kstep(ks + 1, std::integral_constant<int, 1>{});
        }
        if (ks < nk) kstep(ks, std::integral_constant<int, 0>{});
    }
    for (int ks = 0; !PIPE && ks < nk; ++ks) {
        float* cur = DB ? lds + (ks & 1) * STAGE : lds;
        const bool more = ks + 1 < nk;
        if (more) load_step();
        if (BF == 6) {
            const __bf16* Ab = (const __bf16*)cur + (wm * 32 * TM + r) * LDK + h * 8;
            const __bf16* Bb = (const __bf16*)cur + BM * LDK + (wn * 32 * TN + r) * LDK + h * 8;
#pragma unroll
            for (int k16 = 0; k16 < BK / 16; ++k16) {
                bf16x8 a3[TM][3], b3[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 3; ++q) a3[i][q] = *(const bf16x8*)(Ab + i * 32 * LDK + q * BK + k16 * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 3; ++q) b3[j][q] = *(const bf16x8*)(Bb + j * 32 * LDK + q * BK + k16 * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {          // smallest terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][1], b3[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][2], b3[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][0], b3[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][1], b3[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][0], b3[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][0], b3[j][0], acc[i][j], 0, 0, 0);
                    }
            }
        } else if (BF == 3) {
            const __bf16* Ab = (const __bf16*)cur + (wm * 32 * TM + r) * LDK + h * 8;
            const __bf16* Bb = (const __bf16*)cur + BM * LDK + (wn * 32 * TN + r) * LDK + h * 8;
#pragma unroll
            for (int k16 = 0; k16 < BK / 16; ++k16) {
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *(const bf16x8*)(Ab + i * 32 * LDK + k16 * 16);
                    al[i] = *(const bf16x8*)(Ab + i * 32 * LDK + BK + k16 * 16);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *(const bf16x8*)(Bb + j * 32 * LDK + k16 * 16);
                    bl[j] = *(const bf16x8*)(Bb + j * 32 * LDK + BK + k16 * 16);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {          // small terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else if (BF == 2) {
            // v_mfma_f32_32x32x16_f16: same fragment layout as the bf16 form
            const _Float16* Ab = (const _Float16*)cur + (wm * 32 * TM + r) * LDK + h * 8;
            const _Float16* Bb = (const _Float16*)cur + BM * LDK + (wn * 32 * TN + r) * LDK + h * 8;
#pragma unroll
            for (int k16 = 0; k16 < BK / 16; ++k16) {
                f16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *(const f16x8*)(Ab + i * 32 * LDK + k16 * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *(const f16x8*)(Bb + j * 32 * LDK + k16 * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else if (BF) {
            // v_mfma_f32_32x32x16_bf16: lane (r, h) feeds A[row r][k = 8h .. 8h+7] and B[k = 8h .. 8h+7][col r]: 16 bytes each
            const __bf16* Ab = (const __bf16*)cur + (wm * 32 * TM + r) * LDK + h * 8;
            const __bf16* Bb = (const __bf16*)cur + BM * LDK + (wn * 32 * TN + r) * LDK + h * 8;
#pragma unroll
            for (int k16 = 0; k16 < BK / 16; ++k16) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8*)(Ab + i * 32 * LDK + k16 * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8*)(Bb + j * 32 * LDK + k16 * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
        const float* As = cur + (wm * 32 * TM + r) * LDK + h * 4;
        const float* Bs = cur + BM * LDK + (wn * 32 * TN + r) * LDK + h * 4;
        // fragments of chunk kk+1 are requested before the MFMAs of chunk kk are issued (LDS latency under matrix work)
        f32x4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = *(const f32x4*)(As + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = *(const f32x4*)(Bs + j * 32 * LDK);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int cu = kk & 1, nx = cu ^ 1;
            if (kk + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nx][i] = *(const f32x4*)(As + i * 32 * LDK + (kk + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nx][j] = *(const f32x4*)(Bs + j * 32 * LDK + (kk + 1) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);       // keep the reads one chunk ahead (hipcc would sink them to first use)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[cu][i][e], b[cu][j][e], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (DB) {
            if (more) store_step(lds + ((ks + 1) & 1) * STAGE);
            __syncthreads();
        } else {
            __syncthreads();                   // every wave is done reading the stage
            if (more) store_step(lds);
            __syncthreads();
        }
    }

    IG_STAMP(2);
#if IGEMM_ABL & 1      // ablation (tools/micro/igemm_ablate.hip): no epilogue at all
    {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
        if (sum == 123.456f) p.out[0] = sum;
        return;
    }
#endif
    // ---- direct epilogue (stride-1 gathers, whole tiles): every accumulator register goes straight to HBM.  Register e
    // of lane (r, h) is row (e & 3) + 8 (e >> 2) + 4 h, column r of its 32 x 32 tile, so one wave store instruction writes two
    // rows x 128 contiguous bytes (whole lines).  No LDS transpose, no barrier, a quarter of the instructions of the LDS
    // epilogue below -- and instruction ISSUE outside the K loop is what a tile pays for: while this workgroup is in its
    // prologue / epilogue the co-resident one streams MFMAs and this one only advances in the gaps of that stream
    // (profiles/r03_igemm_phases.md: 10 k cycles of epilogue alone, 33-40 k next to a streaming partner). ----
    {
    int64_t R0 = m0, RS = 1, limit = p.M - m0;  // output row of tile row lr = R0 + lr * RS; tile rows < limit exist
    if (POS) {
        limit = p.N - m0;
        if (p.hwnc) { R0 = (int64_t)pos * p.N + m0; RS = 1; }
        else { R0 = m0 * HoWo + pos; RS = HoWo; }
    }
    // whole tiles only (every row and column of the tile exists: all but the last row / column tile of a launch), so that the
    // path is straight-line code without a single per-lane predicate; ragged tiles take the LDS epilogue below
    if (IGEMM_EPI && !IO && TS == 1 && limit >= BM && n0 + BN <= p.Cout) {
        const int64_t pitch = RS * p.Cout;          // floats between consecutive tile rows (workgroup-uniform)
        const int lrow0 = wm * 32 * TM + 4 * h;     // this lane's tile row for e = 0, i = 0
        const int colb = n0 + wn * 32 * TN + r;     // ... and its column for j = 0
        const int64_t lane_off = (R0 + (int64_t)lrow0 * RS) * p.Cout + colb;
        float* outp = p.out + lane_off;
        float scl[TN], sft[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            scl[j] = p.scale ? p.scale[colb + 32 * j] : 1.f;
            sft[j] = p.shift ? p.shift[colb + 32 * j] : 0.f;
        }
        double st0[TN], st1[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) st0[j] = st1[j] = 0.0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float res[16][TN];
            if (p.residual) {                       // all residual loads of this row-tile in flight before any is used
                const float* resp = p.residual + lane_off;
#pragma unroll
                for (int e = 0; e < 16; ++e)
#pragma unroll
                    for (int j = 0; j < TN; ++j) res[e][j] = resp[(i * 32 + (e & 3) + 8 * (e >> 2)) * pitch + 32 * j];
                if (p.res_mask) {                   // one byte per channel quad of a row: bit (column & 3) = keep
                    const uint8_t* mp = p.res_mask;
                    unsigned mk[16][TN];
#pragma unroll
                    for (int e = 0; e < 16; ++e)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            mk[e][j] = mp[(lane_off + (i * 32 + (e & 3) + 8 * (e >> 2)) * pitch + 32 * j) >> 2];
#pragma unroll
                    for (int e = 0; e < 16; ++e)
#pragma unroll
                        for (int j = 0; j < TN; ++j) res[e][j] = (mk[e][j] >> (r & 3)) & 1u ? res[e][j] : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
#pragma unroll
                    for (int j = 0; j < TN; ++j) res[e][j] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float a = acc[i][j][e];
                    if (p.stats) { st0[j] += (double)a; st1[j] += (double)a * (double)a; }
                    const float x = a * scl[j] + sft[j] + res[e][j];           // the LDS epilogue's expression
                    outp[(i * 32 + (e & 3) + 8 * (e >> 2)) * pitch + 32 * j] = p.relu ? fmaxf(x, 0.f) : x;
                }
            }
        }
        if (p.stats) {
            // column sums: the lane's 32 rows, its half-wave partner's 32 (fixed xor), then the row-waves in order through LDS
            // (the stages are dead: the K loop ended on a barrier)
            constexpr int WM = BM / (32 * TM);
            double* S = (double*)lds;                       // [WM][2][BN]
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                st0[j] += __shfl_xor(st0[j], 32);
                st1[j] += __shfl_xor(st1[j], 32);
                if (h == 0) {
                    S[(wm * 2 + 0) * BN + wn * 32 * TN + j * 32 + r] = st0[j];
                    S[(wm * 2 + 1) * BN + wn * 32 * TN + j * 32 + r] = st1[j];
                }
            }
            __syncthreads();
            for (int u = tid; u < 2 * BN; u += NT) {            // 2 BN sums, NT threads (BN = 256: two per thread)
                const int which = u / BN, cc = u % BN;
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < WM; ++q) t += S[(q * 2 + which) * BN + cc];
                p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + n0 + cc] = t;
            }
        }
        IG_STAMP(3);
        return;
    }
    }
    // ---- direct epilogue of the transposed gather (TS == 2, round 4): the rows of a tile are the pixels of one parity class, so they
    // are not equally spaced -- their output offsets go through a 1 KB table in LDS (one decode per row instead of one per element)
    // and every accumulator register is stored straight from there, as above.  A stride-2 input-gradient tile is 4-32 K-steps
    // between a prologue and an epilogue (one to four filter taps): the LDS-transpose epilogue (~10 k cycles alone, 33-40 k beside a
    // streaming partner) was a tenth to a third of such a workgroup. ----
    if constexpr (TS > 1) {
        if (IGEMM_EPI && !IO && !p.stats && Mc - m0 >= BM && n0 + BN <= p.Cout && !p.scale && !p.shift && !p.relu) {
            int64_t* rowoff = (int64_t*)lds;                    // the stages are dead: the K loop ended on a barrier
            for (int lr = tid; lr < BM; lr += NT) {
                int64_t n;
                int oy, ox;
                decode_row(m0 + lr, n, oy, ox);
                rowoff[lr] = ((n * p.Ho + oy) * p.Wo + ox) * (int64_t)p.Cout;
            }
            __syncthreads();
            const int colb = n0 + wn * 32 * TN + r;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int64_t off[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) off[e] = rowoff[wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] + colb;
                float res[16][TN];
                if (p.residual) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
#pragma unroll
                        for (int j = 0; j < TN; ++j) res[e][j] = p.residual[off[e] + 32 * j];
                    if (p.res_mask) {
                        unsigned mk[16][TN];             // all mask bytes requested before the first select (see the LDS epilogue below)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
#pragma unroll
                            for (int j = 0; j < TN; ++j) mk[e][j] = p.res_mask[(off[e] + 32 * j) >> 2];
#pragma unroll
                        for (int e = 0; e < 16; ++e)
#pragma unroll
                            for (int j = 0; j < TN; ++j) res[e][j] = (mk[e][j] >> (r & 3)) & 1u ? res[e][j] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
#pragma unroll
                        for (int j = 0; j < TN; ++j) res[e][j] = 0.f;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e)
#pragma unroll
                    for (int j = 0; j < TN; ++j) p.out[off[e] + 32 * j] = acc[i][j][e] * 1.f + 0.f + res[e][j];   // the LDS epilogue's expression (scale 1, shift 0)
            }
            IG_STAMP(3);
            return;
        }
    }
    // ---- epilogue: accumulators -> LDS tile [BM/TM][BN+4] -> 16-byte pieces of contiguous output rows; one pass per
    // accumulator row-tile so the tile never needs more LDS than the K-loop stages ----
    float* C = lds;
    constexpr int EM = BM / TM;                // rows per epilogue pass: wave wm contributes rows [wm*32, wm*32+32)
    constexpr int C4 = BN / 4;                 // 16-byte pieces per tile row
    constexpr int RP = NT / C4;                // tile rows covered per sweep
    constexpr int NPASS = EM / RP;
    const int c4 = tid % C4, rr = tid / C4;
    const int col = n0 + c4 * 4;
    const bool aligned = (p.Cout & 3) == 0;    // otherwise rows are not 16-byte aligned (odd-sized k-NN bank): element-wise
    const bool col_ok = col < p.Cout;
    // train-mode BatchNorm statistics taken while the tile is in LDS (raw accumulators; rows past M are exact zeros):
    // thread -> one column, one slice of the pass's rows, summed in double in a fixed order
    constexpr int SL = NT / BN, RS = EM / SL;
    const int scol = tid % BN, ssl = tid / BN;
    double st0 = 0.0, st1 = 0.0;
    f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
    if (aligned && col_ok && p.scale) s4 = *(const f32x4*)(p.scale + col);
    if (aligned && col_ok && p.shift) t4 = *(const f32x4*)(p.shift + col);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (i) __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                C[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + (wn * TN + j) * 32 + r] = acc[i][j][e];
        __syncthreads();
        if (p.stats) {
            for (int rw = 0; rw < RS; ++rw) {
                const double v = (double)stored<io_t>(C[(ssl * RS + rw) * LDC + scol]);
                st0 += v;
                st1 += v * v;
            }
        }
        int64_t o[NPASS];
        f32x4 res[NPASS];
        f16x4 rraw[NPASS];
        unsigned mkq[NPASS];
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const int lr = rr + q * RP;                                      // row inside the pass tile
            int64_t row = m0 + ((lr >> 5) * TM + i) * 32 + (lr & 31);        // lr>>5 = the wave-row that produced it
            bool ok = col_ok;
            if (POS) {
                ok = ok && row < p.N;
                row = p.hwnc ? (int64_t)pos * p.N + row : row * HoWo + pos;
            } else if (TS > 1) {
                int64_t n;
                int oy, ox;
                ok = decode_row(row, n, oy, ox) && ok;
                row = (n * p.Ho + oy) * p.Wo + ox;
            } else {
                ok = ok && row < p.M;
            }
            o[q] = ok ? row * p.Cout + col : -1;
            // every pass's residual piece (and mask byte) is REQUESTED here; conversions and masking follow the loop.  With the half
            // conversion / the mask select inside this loop hipcc waited for each piece before asking for the next (round 6: NPASS L2
            // round trips in a row in the epilogue of the masked and the half-tensor input gradients)
            if (IO) rraw[q] = *(const f16x4*)((aligned && ok && p.residual) ? (const void*)((const hf*)p.residual + o[q]) : (const void*)g_zero_page);
            else res[q] = *(const f32x4*)((aligned && ok && p.residual) ? p.residual + o[q] : g_zero_page);
            mkq[q] = 0xfu;
            if (aligned && ok && p.res_mask) mkq[q] = p.res_mask[o[q] >> 2];
        }
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            if (IO) res[q] = __builtin_convertvector(rraw[q], f32x4);
            if (p.res_mask) {
#pragma unroll
                for (int k = 0; k < 4; ++k) res[q][k] = (mkq[q] >> k) & 1u ? res[q][k] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            f32x4 v = *(const f32x4*)(C + (rr + q * RP) * LDC + c4 * 4);
            if (aligned) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float x = v[k] * s4[k] + t4[k] + res[q][k];
                    v[k] = p.relu ? fmaxf(x, 0.f) : x;
                }
                if (o[q] >= 0) {
                    if (IO) st4((hf*)p.out + o[q], v);
                    else *(f32x4*)(p.out + o[q]) = v;
                }
            } else if (o[q] >= 0) {
                for (int k = 0; k < 4 && col + k < p.Cout; ++k) {
                    float x = v[k] * (p.scale ? p.scale[col + k] : 1.f) + (p.shift ? p.shift[col + k] : 0.f);
                    if (p.residual) x += p.residual[o[q] + k];
                    p.out[o[q] + k] = p.relu ? fmaxf(x, 0.f) : x;
                }
            }
        }
    }
    if (p.stats) {
        __syncthreads();
        double* S = (double*)lds;                       // [SL][2][BN]
        S[(ssl * 2 + 0) * BN + scol] = st0;
        S[(ssl * 2 + 1) * BN + scol] = st1;
        __syncthreads();
        for (int u = tid; u < 2 * BN; u += NT) {
            const int which = u / BN, cc = u % BN;
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < SL; ++q) t += S[(q * 2 + which) * BN + cc];
            if (n0 + cc < p.Cout) p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + n0 + cc] = t;
        }
    }
    IG_STAMP(3);
}

// Position order for the position-major kernels: heaviest (most in-bounds taps) first, see the kernel.
static void sort_positions(ConvParams& p, int sample_groups) {
    // 0: engine-stream mapping (round 1); 1: heaviest position first over chunks of SSAD_POS_CHUNK sample groups (rounds 2-3);
    // 2: the same order inside XCD-local blocks of a few sample groups (round 4)
    static const int mode = getenv("SSAD_POS_LPT") ? atoi(getenv("SSAD_POS_LPT")) : 1;
    const int HoWo = p.Ho * p.Wo;
    // chunk of sample groups swept position by position: 32 groups x 128 samples = 4096 maps, 134 MB of layer2 input -- inside the
    // 256 MB Infinity Cache.  Same speed as one chunk (654 maps/s either way), 0 = one chunk
    static const int chunk = getenv("SSAD_POS_CHUNK") ? atoi(getenv("SSAD_POS_CHUNK")) : 32;
    static const int xcd_wgs = getenv("SSAD_POS_XCD_WGS") ? atoi(getenv("SSAD_POS_XCD_WGS")) : 64;
    p.pos_lpt = 0;
    p.pos_sg = sample_groups;
    p.pos_chunk = chunk > 0 && chunk < sample_groups ? chunk : sample_groups;
    if (!mode || HoWo > 64 || HoWo < 2) return;
    int taps[64], order[64];
    bool uniform = true;
    for (int pos = 0; pos < HoWo; ++pos) {
        const int oy = pos / p.Wo, ox = pos - oy * p.Wo;
        int ny = 0, nx = 0;
        for (int k = 0; k < p.KH; ++k) ny += (unsigned)(oy * p.stride - p.pad + k) < (unsigned)p.H;
        for (int k = 0; k < p.KW; ++k) nx += (unsigned)(ox * p.stride - p.pad + k) < (unsigned)p.W;
        taps[pos] = ny * nx;
        order[pos] = pos;
        uniform = uniform && taps[pos] == taps[0];
    }
    if (uniform && mode != 2) return;                    // nothing to balance: keep the engine-stream mapping
    for (int i = 1; i < HoWo; ++i)                       // stable insertion sort, heaviest first
        for (int j = i; j > 0 && taps[order[j]] > taps[order[j - 1]]; --j) { int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    for (int i = 0; i < HoWo; ++i) p.pos_tab[i] = (unsigned char)order[i];
    p.pos_lpt = 1;
    if (mode == 2 && sample_groups >= 64) {
        p.pos_lpt = 2;
        p.pos_chunk = xcd_wgs / HoWo > 0 ? xcd_wgs / HoWo : 1;      // sample groups per XCD-local block: ~one round of resident workgroups
    }
}

template <int BM, int BN, int TM, int TN, int BK, int TS, bool POS, bool DB = true, int BF = 0, bool IO = false>
int launch(const ConvParams& p, hipStream_t st) {
    constexpr int stage_bytes = (DB ? 2 : 1) * (BM + BN) *
                                (BF == 6 ? (3 * BK + 8) * 2 : (BF == 3 ? (2 * BK + 8) * 2 : (BF ? (BK + 8) * 2 : (BK + 4) * 4)));
    constexpr int epi_bytes = (BM / TM) * (BN + 4) * 4;
    constexpr int lds_min = (stage_bytes > epi_bytes ? stage_bytes : epi_bytes) + ((IGEMM_ABL & 64) ? 40 * 1024 : 0);
    // SSAD_CONV_LDS_PAD_<BN>: extra LDS bytes per workgroup = fewer resident workgroups (launch-quantisation experiments)
    static const int lds_pad = getenv(BN == 64 ? "SSAD_CONV_LDS_PAD_64" : "SSAD_CONV_LDS_PAD_128")
                                   ? atoi(getenv(BN == 64 ? "SSAD_CONV_LDS_PAD_64" : "SSAD_CONV_LDS_PAD_128")) : 0;
    const int lds_bytes = lds_min + lds_pad;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_f32_kernel<BM, BN, TM, TN, BK, TS, POS, DB, BF, IO>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        attr_set = true;
    }
    ConvParams q = p;
    q.pos_lpt = 0;
    if (POS) sort_positions(q, (int)cdiv64(p.N, BM));
    int64_t gx = POS ? (q.pos_lpt == 2 ? cdiv64(cdiv64(q.pos_sg, 8), q.pos_chunk) * q.pos_chunk * p.Ho * p.Wo * 8
                        : q.pos_lpt ? cdiv64(q.pos_sg, q.pos_chunk) * q.pos_chunk * p.Ho * p.Wo
                                    : cdiv64(cdiv64(p.N, BM), 32) * 32 * p.Ho * p.Wo)
                     : cdiv64(p.M, BM);
    if (TS > 1) {
        // parity classes ordered by the taps they keep, most first: the launch then ends on its lightest workgroups (measured,
        // batch 256: 3 x 3 dgrads 423 / 356 / 403 -> 361 / 347 / 394 us; lightest-first 1 x 1: 85 -> 117 us, hence the sort)
        int taps[4];
        for (int c = 0; c < 4; ++c) {
            taps[c] = 0;
            for (int ky = 0; ky < p.KH; ++ky)
                for (int kx = 0; kx < p.KW; ++kx)
                    taps[c] += (((c >> 1) - p.pad + ky) & 1) == 0 && (((c & 1) - p.pad + kx) & 1) == 0;
            q.cls_id[c] = c;
        }
        for (int i = 1; i < 4; ++i)
            for (int j = i; j > 0 && taps[q.cls_id[j]] > taps[q.cls_id[j - 1]]; --j) { int t = q.cls_id[j]; q.cls_id[j] = q.cls_id[j - 1]; q.cls_id[j - 1] = t; }
        gx = 0;
        for (int c = 0; c < 4; ++c) {
            q.cls_start[c] = (int)gx;
            gx += cdiv64(p.N * ((p.Ho - (q.cls_id[c] >> 1) + 1) / 2) * ((p.Wo - (q.cls_id[c] & 1) + 1) / 2), BM);
        }
        q.cls_start[4] = (int)gx;
    }
    dim3 grid((unsigned)gx, (unsigned)((p.Cout + BN - 1) / BN));
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, TM, TN, BK, TS, POS, DB, BF, IO>), grid,
                       dim3((BM / (32 * TM)) * (BN / (32 * TN)) * 64), lds_bytes, st, q);
    return (int)gx;
}

// every dispatch returns the number of row workgroups launched (= rows of the statistics partials)
template <int TS, int BF = 1, bool IO = false>
int dispatch_bf16(const ConvParams& p, hipStream_t st) {
    if (p.Cout <= 64) return launch<256, 64, 2, 2, 32, TS, false, true, BF, IO>(p, st);
    return launch<128, 128, 2, 2, 32, TS, false, true, BF, IO>(p, st);
}

// three-way split ("bf16x6"): rows are 1.5x the fp32 bytes.  Measured: one LDS stage with BK = 32 (53 KB, three
// workgroups per CU, 48 MFMAs per wave between barriers) beats two stages with BK = 16 by 7-10 %
template <int TS, bool POS>
int dispatch_x6(const ConvParams& p, hipStream_t st) {
    static const int v = getenv("SSAD_X6_VARIANT") ? atoi(getenv("SSAD_X6_VARIANT")) : 1;
    if (v == 1) {
        if (p.Cout <= 64) return launch<256, 64, 2, 2, 32, TS, POS, false, 6>(p, st);
        return launch<128, 128, 2, 2, 32, TS, POS, false, 6>(p, st);
    }
    if (p.Cout <= 64) return launch<256, 64, 2, 2, 16, TS, POS, true, 6>(p, st);
    return launch<128, 128, 2, 2, 16, TS, POS, true, 6>(p, st);
}

// split-bf16 ("bf16x3") form: same LDS bytes per row as fp32
template <int TS, bool POS>
int dispatch_x3(const ConvParams& p, hipStream_t st) {
    // measured: one LDS stage with BK = 32 (37-46 KB, three workgroups per CU) is 6-16 % faster than the exact kernel's
    // double-buffered tiles -- with 3/16 of the MFMA time the stage barrier matters less than residency
    static const int v = getenv("SSAD_X3_VARIANT") ? atoi(getenv("SSAD_X3_VARIANT")) : 1;
    if (v == 1) {
        if (p.Cout <= 64) return launch<256, 64, 2, 2, 32, TS, POS, false, 3>(p, st);
        return launch<128, 128, 2, 2, 32, TS, POS, false, 3>(p, st);
    }
    if (p.Cout <= 64) return launch<256, 64, 2, 2, 16, TS, POS, true, 3>(p, st);
    return launch<128, 128, 2, 2, 32, TS, POS, true, 3>(p, st);
}

// Tile of the exact-fp32 instantiation a problem is given (also reported by ssad_conv_igemm_tile: bench.py names the
// instantiations its roofline sums over)
enum IgemmTile { T_256x64_K16 = 0, T_256x64_SB, T_128x64, T_256x128, T_256x128_W4, T_128x256, T_64x64, T_128x128, T_256x256, T_128x64_SB, T_128x64_K16, T_128x128_K16, T_128x256_K16 };

template <bool POS>
IgemmTile pick_tile(const ConvParams& p) {
    static const int variant = getenv("SSAD_CONV64_VARIANT") ? atoi(getenv("SSAD_CONV64_VARIANT")) : 1;
    // ring launches (a skipped square of positions: the patch-scoring pass's layer1, round 6): what is left are the border positions,
    // 8-18 K-steps per workgroup -- the 128-row tile (twice the workgroups, half the prologue each) measured 314 against 320 ms per
    // 256 images for the pass's position-major convs.  With 16-float K-steps its two LDS stages take 30 KB instead of 55: FOUR workgroups
    // per CU instead of two, so that a prologue / epilogue (10-25 k + 7-16 k cycles beside 24-55 k of K loop, tools/micro/igemm_var.hip
    // with -DIGEMM_TRACE) finds other workgroups' matrix loops to hide behind: 2.58 -> 2.47 ms per launch (SSAD_CONV_RING_VARIANT=3: the
    // 32-float form, 1: one stage)
    static const int ring = getenv("SSAD_CONV_RING_VARIANT") ? atoi(getenv("SSAD_CONV_RING_VARIANT")) : 0;
    if (p.Cout <= 64 && POS && p.skip_lo <= p.skip_hi && !getenv("SSAD_CONV64_VARIANT"))
        return ring == 1 ? T_128x64_SB : ring == 3 ? T_128x64 : T_128x64_K16;
    if (p.Cout <= 64) return variant == 2 ? T_256x64_SB : variant == 1 ? T_256x64_K16 : T_128x64;
    static const int big = getenv("SSAD_CONV128_VARIANT") ? atoi(getenv("SSAD_CONV128_VARIANT")) : 0;
    if (big == 7) return T_128x128_K16;                                  // 16-float K-steps: 41 KB of LDS, three workgroups per CU
    if (big == 1) return T_256x128;
    if (big == 2) return T_256x128_W4;                                    // one workgroup per CU, four waves of 128 x 64
    // 256 x 256 tile, four waves of 128 x 128 (16 accumulator tiles each): 16 staged pieces per 256 MFMAs (SSAD_CONV128_VARIANT=6,
    // position-major launches with Cout % 256 == 0 only: round-4 experiment)
    if (big == 6 && POS && p.Cout % 256 == 0 && cdiv64(p.N, 256) * p.Ho * p.Wo * (p.Cout / 256) >= 256) return T_256x256;
    // 128 x 256 tile (four waves of 64 x 128, one workgroup per CU) where Cout is a multiple of 256 and the launch still fills the
    // chip: 12 staged pieces per 128 MFMAs instead of 16 -- the per-piece cost is what a K-step loses (profiles/r03_igemm_phases.md).
    // Measured (same run): layer3 / layer4 training convs 0.635 -> 0.612 / 0.606 -> 0.586 ms, position-major scoring convs
    // 1.649 -> 1.605 / 1.056 -> 1.017 ms; batch-32 grids (64-128 workgroups) would lose 3-5x, hence the floor.
    static const int wide_min = getenv("SSAD_CONV_WIDE_GRID") ? atoi(getenv("SSAD_CONV_WIDE_GRID")) : 256;
    if (big != 5 && p.Cout % 256 == 0 && wide_min > 0) {
        const int64_t rows = POS ? cdiv64(p.N, 128) * p.Ho * p.Wo : cdiv64(p.M, 128);
        // ... with 16-float K-steps (61 KB of LDS, 256 registers: the staging sets halve) TWO such workgroups fit a CU and one's prologue
        // and epilogue hide behind the other's K loop: 9.79 -> 9.51 / 6.29 -> 6.15 ms on the layer3 / layer4 shapes of the scoring pass
        // (SSAD_CONV256_K16: 0 off, 1 position-major launches only, 2 all -- the WideResNet-50 1 x 1 convs gain 0.5 %, the training step nothing)
        static const int wide_k16 = getenv("SSAD_CONV256_K16") ? atoi(getenv("SSAD_CONV256_K16")) : 2;
        if (rows * (p.Cout / 256) >= wide_min) return (wide_k16 == 2 || (POS && wide_k16)) ? T_128x256_K16 : T_128x256;
    }
    // small problems (batch 32-96 on the 8x8 / 16x16 maps of layer3 / layer4): 128x128 tiles leave CUs idle -- fewer than
    // ~1.5 workgroups per CU -- so the 128x64 tile doubles the grid (measured at batch 96 / 32: see DESIGN.md)
    static const int small_min = getenv("SSAD_CONV_SMALL_GRID") ? atoi(getenv("SSAD_CONV_SMALL_GRID")) : 500;
    // ... and when even that leaves most CUs without a workgroup (batch 32 on the 8x8 maps of layer4: 128 workgroups of
    // 144 K-steps each) the 64x64 tile doubles the grid again (measured at batch 32: layer4 3x3 0.204 -> see DESIGN.md)
    // (round 6, with the buffer-load K loop: 520 instead of 200 -- layer3 at batch 32 and layer4 at batch 96 take the 64 x 64 tile too:
    // batch 32 5.265 -> 5.20 ms, batch 96 13.29 -> 12.86 ms, batch 256 untouched)
    static const int tiny_min = getenv("SSAD_CONV_TINY_GRID") ? atoi(getenv("SSAD_CONV_TINY_GRID")) : 520;
    if (!POS) {
        const int64_t g128 = cdiv64(p.M, 128) * ((p.Cout + 127) / 128);
        if (cdiv64(p.M, 128) * ((p.Cout + 63) / 64) < tiny_min && p.M > 128) return T_64x64;
        if (g128 < small_min) return T_128x64;
    }
    // position-major launches: 16-float K-steps (41 KB of LDS: three workgroups per CU instead of two), 11.70 -> 11.53 ms on the
    // layer2 shape of the scoring pass; the pixel-major training launches measured the same either way
    static const int k16 = getenv("SSAD_CONV128_K16") ? atoi(getenv("SSAD_CONV128_K16")) : 1;
    if (POS && k16) return T_128x128_K16;
    return T_128x128;
}

template <int TS, bool POS>
int dispatch(const ConvParams& p, hipStream_t st) {
    switch (pick_tile<POS>(p)) {
        case T_256x64_SB: return launch<256, 64, 2, 2, 32, TS, POS, false>(p, st);
        case T_256x64_K16: return launch<256, 64, 2, 2, 16, TS, POS>(p, st);
        case T_128x64: return launch<128, 64, 1, 2, 32, TS, POS>(p, st);
        case T_128x64_SB: return launch<128, 64, 1, 2, 32, TS, POS, false>(p, st);      // one LDS stage: 27 KB, up to five workgroups per CU
        case T_128x64_K16: return launch<128, 64, 1, 2, 16, TS, POS>(p, st);            // 16-float K-steps: 30 KB, two stages
        case T_128x128_K16: return launch<128, 128, 2, 2, 16, TS, POS>(p, st);
        case T_128x256_K16: return launch<128, 256, 2, 4, 16, TS, POS>(p, st);          // 61 KB, 256 registers: two workgroups per CU
        case T_256x128: return launch<256, 128, 2, 2, 32, TS, POS>(p, st);
        case T_256x128_W4: return launch<256, 128, 4, 2, 32, TS, POS>(p, st);
        case T_128x256: return launch<128, 256, 2, 4, 32, TS, POS>(p, st);
        case T_64x64: return launch<64, 64, 1, 1, 32, TS, POS>(p, st);
        case T_256x256:
            if constexpr (POS && TS == 1) return launch<256, 256, 4, 4, 32, TS, POS>(p, st);
            else return launch<128, 128, 2, 2, 32, TS, POS>(p, st);
        default: return launch<128, 128, 2, 2, 32, TS, POS>(p, st);
    }
}

// buffer-load staging (IGEMM_BUFLD): the rows of a tile -- up to 256 consecutive output pixels, i.e. 256 / (Ho Wo) + 2 input images, or
// 256 samples of a position-major launch -- are addressed by 32-bit byte offsets from the tile's first image
static bool tile_span_ok(const ConvParams& p, bool posmajor, int elt_bytes, int ts = 1) {
    const int64_t image = (int64_t)p.H * p.W * p.Cin * elt_bytes;
    if (posmajor) return p.hwnc || 256 * image < ((int64_t)1 << 31);
    const int64_t rows_per_image = ts > 1 ? (int64_t)(p.Ho / 2) * (p.Wo / 2) : (int64_t)p.Ho * p.Wo;      // (a parity class of the transposed gather)
    return (256 / (rows_per_image > 0 ? rows_per_image : 1) + 2) * image < ((int64_t)1 << 31);
}

int conv_fwd_impl(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                  const float* residual, int relu, int64_t N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                  int pad, int hwnc, void* stream, int bf16 = 0, double* stats = nullptr, int* stat_rows = nullptr, int io16 = 0,
                  int skip_lo = 1, int skip_hi = 0) {
    SSAD_CHECK_ARG(in && w_ohwi && out, "null pointer");
    SSAD_CHECK_ARG(!io16 || (bf16 == 2 && !hwnc && Cout % 4 == 0), "half tensors: fp16 operands, NHWC, Cout % 4 == 0");
    SSAD_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "empty shape");
    if ((bf16 == 0 || bf16 == 1 || bf16 == 2) && !io16 && !hwnc && KH == 1 && KW == 1 && H == 1 && W == 1 && stride == 1 && pad == 0 &&
        ssad_linear_small_ok(in, w_ohwi, N, Cin))          // a linear layer over a training batch's few rows (16-bit modes: operands
                                                           // rounded while loaded, the same arithmetic as the 16-bit MFMA)
        return ssad_linear_small_launch(in, w_ohwi, out, scale, shift, residual, relu, (int)N, Cin, Cout, stats, stat_rows, stream, bf16);
    SSAD_CHECK_ARG(Cin % KALIGN == 0, "Cin must be a multiple of 32");
    SSAD_CHECK_ARG(KH > 0 && KW > 0 && stride > 0 && pad >= 0 && KH * KW <= 32, "bad filter geometry (<= 32 taps)");
    ConvParams p;
    p.pos_lpt = 0; p.pos_sg = 0; p.pos_chunk = 1;
    p.skip_lo = skip_lo; p.skip_hi = skip_hi;
    SSAD_CHECK_ARG(skip_lo > skip_hi || hwnc, "a skipped rectangle of output positions: position-major tensors only");
    p.in = in; p.wt = w_ohwi; p.out = out; p.scale = scale; p.shift = shift; p.residual = residual; p.res_mask = nullptr;
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.relu = relu;
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    SSAD_CHECK_ARG(p.Ho > 0 && p.Wo > 0, "empty output");
    p.M = N * p.Ho * p.Wo;
    p.N = N;
    p.K = KH * KW * Cin;
    p.hwnc = hwnc;
    p.stats = stats;
    // position-major rows pay off when padding is a visible share of the taps (small maps, many samples); in the
    // [H][W][N][C] layout they are also what makes a workgroup's rows contiguous
    const bool posmajor = hwnc || (!stats && pad > 0 && N >= 128 && p.Ho * p.Wo <= 4);
    SSAD_CHECK_ARG(cdiv64(p.M, 128) + 32 * p.Ho * p.Wo < (int64_t)2147483647, "M too large for one launch");
    SSAD_CHECK_ARG(tile_span_ok(p, posmajor, io16 ? 2 : 4), "input images too large: the rows of a tile must span less than 2 GB");
    hipStream_t st = (hipStream_t)stream;
    int rows;
    SSAD_CHECK_ARG(bf16 == 0 || bf16 == 1 || bf16 == 2 || bf16 == 3 || bf16 == 6,
                   "operand mode: 0 (fp32), 1 (bf16), 2 (fp16), 3 (bf16x3), 6 (bf16x6)");
    if (bf16 == 6) {
        rows = posmajor ? dispatch_x6<1, true>(p, st) : dispatch_x6<1, false>(p, st);
    } else if (bf16 == 3) {
        rows = posmajor ? dispatch_x3<1, true>(p, st) : dispatch_x3<1, false>(p, st);
    } else if (bf16) {
        SSAD_CHECK_ARG(!hwnc, "16-bit operands: NHWC only");
        rows = io16 ? dispatch_bf16<1, 2, true>(p, st) : bf16 == 2 ? dispatch_bf16<1, 2>(p, st) : dispatch_bf16<1, 1>(p, st);
    } else if (posmajor) rows = dispatch<1, true>(p, st);
    else rows = dispatch<1, false>(p, st);
    if (stat_rows) *stat_rows = rows;
    SSAD_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// bf16-operand form of ssad_conv_igemm_fwd (fp32 tensors in HBM, operands rounded to bf16 in the loader, fp32
// accumulate): what torch.autocast does to the same Conv2d / Linear call sites under Trainer(precision=16)
// (src/self_supervised/tools.py:263).
extern "C" int ssad_conv_igemm_fwd_bf16(const float* in, const float* w_ohwi, float* out, const float* scale,
                                        const float* shift, const float* residual, int relu, int64_t N, int H, int W,
                                        int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream) {
    return conv_fwd_impl(in, w_ohwi, out, scale, shift, residual, relu, N, H, W, Cin, Cout, KH, KW, stride, pad, 0, stream, 1);
}

// fp16-operand form: the arithmetic of the reference's own training precision (fp16 autocast under
// pl.Trainer(precision=16), src/self_supervised/tools.py:263, :296): operands rounded to fp16 (11-bit significand) while
// staging, v_mfma_f32_32x32x16_f16, fp32 accumulate; tensors stay fp32 in HBM.
extern "C" int ssad_conv_igemm_fwd_f16(const float* in, const float* w_ohwi, float* out, const float* scale,
                                       const float* shift, const float* residual, int relu, int64_t N, int H, int W,
                                       int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream) {
    return conv_fwd_impl(in, w_ohwi, out, scale, shift, residual, relu, N, H, W, Cin, Cout, KH, KW, stride, pad, 0, stream, 2);
}

// Split-bf16 ("bf16x3") form: fp32 tensors in and out, every product formed from (hi, lo) bf16 pairs on the bf16 matrix
// cores, fp32 accumulate.  hwnc != 0 selects the position-major layout of ssad_conv_igemm_fwd_hwnc.
extern "C" int ssad_conv_igemm_fwd_x3(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                                      const float* residual, int relu, int64_t N, int H, int W, int Cin, int Cout, int KH,
                                      int KW, int stride, int pad, int hwnc, void* stream) {
    return conv_fwd_impl(in, w_ohwi, out, scale, shift, residual, relu, N, H, W, Cin, Cout, KH, KW, stride, pad, hwnc, stream, 3);
}

// Three-way split ("bf16x6"): fp32-faithful products (what is dropped is ~2^-25 of each product).
extern "C" int ssad_conv_igemm_fwd_x6(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                                      const float* residual, int relu, int64_t N, int H, int W, int Cin, int Cout, int KH,
                                      int KW, int stride, int pad, int hwnc, void* stream) {
    return conv_fwd_impl(in, w_ohwi, out, scale, shift, residual, relu, N, H, W, Cin, Cout, KH, KW, stride, pad, hwnc, stream, 6);
}

extern "C" int ssad_conv_igemm_fwd(const float* in, const float* w_ohwi, float* out, const float* scale,
                                   const float* shift, const float* residual, int relu, int64_t N, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream) {
    return conv_fwd_impl(in, w_ohwi, out, scale, shift, residual, relu, N, H, W, Cin, Cout, KH, KW, stride, pad, 0, stream);
}

// Convolution (no epilogue) + train-mode BatchNorm statistics of its output in one pass: every workgroup leaves the
// double-precision column sums of its tile in `workspace` (ssad_conv_stats_workspace doubles) and the finalize kernel of
// ssad_bn_stats turns them into mean / invstd / running statistics.  Saves the separate read of z.
extern "C" int64_t ssad_conv_stats_workspace(int64_t N, int Ho, int Wo, int Cout) {
    return cdiv64(N * Ho * Wo, 32) * 2 * Cout;           // 32 = the smallest row tile any kernel uses (linear_small.hip)
}

extern "C" int ssad_conv_igemm_fwd_stats(const float* in, const float* w_ohwi, float* out, int64_t N, int H, int W, int Cin,
                                         int Cout, int KH, int KW, int stride, int pad, int bf16, float eps, float momentum,
                                         float* mean, float* invstd, float* running_mean, float* running_var,
                                         double* workspace, void* stream) {
    SSAD_CHECK_ARG(mean && invstd && workspace, "null pointer");
    int rows = 0;
    int rc = conv_fwd_impl(in, w_ohwi, out, nullptr, nullptr, nullptr, 0, N, H, W, Cin, Cout, KH, KW, stride, pad, 0, stream,
                           bf16, workspace, &rows);
    if (rc) return rc;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    return ssad_bn_finalize_partials(workspace, rows, N * Ho * Wo, Cout, eps, momentum, mean, invstd, running_mean,
                                     running_var, stream);
}

// Same contraction with every activation tensor (in, out, residual) stored position-major, [H][W][N][C].
// This is the layout of the patch-scoring trunk: N = thousands of 64x64 patches whose maps are 16x16 .. 2x2,
// so a workgroup's 128 rows (128 patches at one output position) are contiguous in HBM for every tap and taps
// that fall into the zero padding are skipped as whole K-steps.
extern "C" int ssad_conv_igemm_fwd_hwnc(const float* in, const float* w_ohwi, float* out, const float* scale,
                                        const float* shift, const float* residual, int relu, int64_t N, int H, int W,
                                        int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream) {
    return conv_fwd_impl(in, w_ohwi, out, scale, shift, residual, relu, N, H, W, Cin, Cout, KH, KW, stride, pad, 1, stream);
}

// ssad_conv_igemm_fwd_hwnc over the RING of output positions outside the square skip_lo <= oy, ox <= skip_hi: the positions inside
// are left untouched (ssad_patch_gather_hwnc fills them).  The layer1 convs of the patch-scoring pass (models.py:211-224 in eval
// mode): a patch is a 32 x 32 window of the image at stride 8, so away from the patch's own zero-padded border a conv output is
// the same arithmetic on the same pixels in every patch that covers them -- computed once per image, not once per patch.
extern "C" int ssad_conv_igemm_fwd_hwnc_ring(const float* in, const float* w_ohwi, float* out, const float* scale,
                                             const float* shift, const float* residual, int relu, int64_t N, int H, int W,
                                             int Cin, int Cout, int KH, int KW, int stride, int pad, int skip_lo, int skip_hi,
                                             void* stream) {
    SSAD_CHECK_ARG(skip_lo >= 0 && skip_hi < H && skip_hi < W, "skipped square outside the map");
    return conv_fwd_impl(in, w_ohwi, out, scale, shift, residual, relu, N, H, W, Cin, Cout, KH, KW, stride, pad, 1, stream, 0, nullptr,
                         nullptr, 0, skip_lo, skip_hi);
}

// Which instantiation ssad_conv_igemm_fwd (hwnc = 0) / ssad_conv_igemm_fwd_hwnc (hwnc = 1) runs a problem on, as
// BM * 100000 + BN * 100 + BK, negative when the rows are position-major (POS): what bench.py's roofline object names.
extern "C" int ssad_conv_igemm_tile(int64_t N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int hwnc) {
    ConvParams p;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.skip_lo = hwnc == 2 ? 0 : 1; p.skip_hi = 0;          // hwnc = 2: a ring launch (ssad_conv_igemm_fwd_hwnc_ring)
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    p.M = N * p.Ho * p.Wo;
    const bool posmajor = hwnc || (pad > 0 && N >= 128 && p.Ho * p.Wo <= 4);
    static const int dims[13][3] = {{256, 64, 16}, {256, 64, 32}, {128, 64, 32}, {256, 128, 32}, {256, 128, 32}, {128, 256, 32},
                                    {64, 64, 32}, {128, 128, 32}, {256, 256, 32}, {128, 64, 32}, {128, 64, 16}, {128, 128, 16}, {128, 256, 16}};
    const int t = posmajor ? (int)pick_tile<true>(p) : (int)pick_tile<false>(p);
    const int code = dims[t][0] * 100000 + dims[t][1] * 100 + dims[t][2];
    return posmajor ? -code : code;
}

// dgrad: dx[n][iy][ix][ci] = sum_{ky,kx,co} dy[n][(iy+pad-ky)/s][(ix+pad-kx)/s][co] * w[co][ky][kx][ci] (+ residual).
// w_flipT is ssad_flip_transpose_weight(w): [Cin][KH][KW][Cout] with both taps reversed, so the sum becomes the
// same gather-GEMM with k = (ky', kx', co), numerator row = iy - (KH-1-pad) + ky'.
static int dgrad_impl(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N,
                                     int Hy, int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride,
                                     int pad, void* stream, int bf16, const uint8_t* res_mask = nullptr, int io16 = 0) {
    SSAD_CHECK_ARG(dy && w_flipT && dx, "null pointer");
    SSAD_CHECK_ARG(!io16 || (bf16 == 2 && !res_mask && Cin % 4 == 0), "half tensors: fp16 operands, no residual mask, Cin % 4 == 0");
    SSAD_CHECK_ARG(N > 0 && Hy > 0 && Wy > 0 && Hx > 0 && Wx > 0 && Cin > 0 && Cout > 0, "empty shape");
    if ((bf16 == 0 || bf16 == 1 || bf16 == 2) && !io16 && !res_mask && KH == 1 && KW == 1 && Hy == 1 && Wy == 1 && Hx == 1 && Wx == 1 &&
        stride == 1 && pad == 0 && ssad_linear_small_ok(dy, w_flipT, N, Cout))        // dx[M][Cin] = dy[M][Cout] . w_flipT[Cin][Cout]^T
        return ssad_linear_small_launch(dy, w_flipT, dx, nullptr, nullptr, residual, 0, (int)N, Cout, Cin, nullptr, nullptr, stream, bf16);
    SSAD_CHECK_ARG(Cout % KALIGN == 0, "Cout (the contraction) must be a multiple of 32");
    SSAD_CHECK_ARG(KH > 0 && KW > 0 && pad >= 0 && pad < KH && pad < KW, "bad filter geometry");
    SSAD_CHECK_ARG(stride == 1 || stride == 2, "stride 1 or 2");
    SSAD_CHECK_ARG((Hx + 2 * pad - KH) / stride + 1 == Hy && (Wx + 2 * pad - KW) / stride + 1 == Wy, "dy/dx sizes disagree");
    SSAD_CHECK_ARG(KH == KW && KH * KW <= 32, "square filters with at most 32 taps only");
    ConvParams p;
    p.pos_lpt = 0; p.pos_sg = 0; p.pos_chunk = 1;
    p.skip_lo = 1; p.skip_hi = 0;
    SSAD_CHECK_ARG(!res_mask || (residual && Cin % 4 == 0), "residual mask needs a residual and Cin % 4 == 0");
    p.in = dy; p.wt = w_flipT; p.out = dx; p.scale = nullptr; p.shift = nullptr; p.residual = residual; p.res_mask = res_mask;
    p.H = Hy; p.W = Wy; p.Cin = Cout; p.Cout = Cin; p.KH = KH; p.KW = KW; p.relu = 0;
    p.stride = 1; p.pad = KH - 1 - pad; p.N = N; p.hwnc = 0;
    p.Ho = Hx; p.Wo = Wx; p.stats = nullptr;
    p.M = N * Hx * Wx;
    p.K = KH * KW * Cout;
    SSAD_CHECK_ARG(cdiv64(p.M, 128) < (int64_t)2147483647, "M too large for one launch");
    SSAD_CHECK_ARG(tile_span_ok(p, false, io16 ? 2 : 4, stride), "gradient images too large: the rows of a tile must span less than 2 GB");
    hipStream_t st = (hipStream_t)stream;
    if (bf16 == 6) {
        if (stride == 1) dispatch_x6<1, false>(p, st);
        else dispatch_x6<2, false>(p, st);
    } else if (bf16 == 3) {
        if (stride == 1) dispatch_x3<1, false>(p, st);
        else dispatch_x3<2, false>(p, st);
    } else if (bf16 == 2 && io16) {
        if (stride == 1) dispatch_bf16<1, 2, true>(p, st);
        else dispatch_bf16<2, 2, true>(p, st);
    } else if (bf16 == 2) {
        if (stride == 1) dispatch_bf16<1, 2>(p, st);
        else dispatch_bf16<2, 2>(p, st);
    } else if (bf16) {
        if (stride == 1) dispatch_bf16<1>(p, st);
        else dispatch_bf16<2>(p, st);
    } else if (stride == 1) dispatch<1, false>(p, st);
    else dispatch<2, false>(p, st);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_conv_igemm_dgrad(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N,
                                     int Hy, int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride,
                                     int pad, void* stream) {
    return dgrad_impl(dy, w_flipT, dx, residual, N, Hy, Wy, Cout, Hx, Wx, Cin, KH, KW, stride, pad, stream, 0);
}

// dgrad with the residual taken under a nibble mask (ssad_bn_apply_fwd_mask): dx = dgrad(dy) + residual * mask.  The
// residual is the gradient of a residual block's output and the mask the active set of the block's final ReLU, so the masked
// gradient of the identity branch is formed here instead of being written and re-read as a tensor of its own.
extern "C" int ssad_conv_igemm_dgrad_masked(const float* dy, const float* w_flipT, float* dx, const float* residual,
                                            const uint8_t* res_mask, int64_t N, int Hy, int Wy, int Cout, int Hx, int Wx, int Cin,
                                            int KH, int KW, int stride, int pad, void* stream) {
    return dgrad_impl(dy, w_flipT, dx, residual, N, Hy, Wy, Cout, Hx, Wx, Cin, KH, KW, stride, pad, stream, 0, res_mask);
}

extern "C" int ssad_conv_igemm_dgrad_bf16(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N,
                                          int Hy, int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride,
                                          int pad, void* stream) {
    return dgrad_impl(dy, w_flipT, dx, residual, N, Hy, Wy, Cout, Hx, Wx, Cin, KH, KW, stride, pad, stream, 1);
}

extern "C" int ssad_conv_igemm_dgrad_f16(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N,
                                         int Hy, int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride,
                                         int pad, void* stream) {
    return dgrad_impl(dy, w_flipT, dx, residual, N, Hy, Wy, Cout, Hx, Wx, Cin, KH, KW, stride, pad, stream, 2);
}

extern "C" int ssad_conv_igemm_dgrad_x6(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N,
                                        int Hy, int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride,
                                        int pad, void* stream) {
    return dgrad_impl(dy, w_flipT, dx, residual, N, Hy, Wy, Cout, Hx, Wx, Cin, KH, KW, stride, pad, stream, 6);
}

extern "C" int ssad_conv_igemm_dgrad_x3(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N,
                                        int Hy, int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride,
                                        int pad, void* stream) {
    return dgrad_impl(dy, w_flipT, dx, residual, N, Hy, Wy, Cout, Hx, Wx, Cin, KH, KW, stride, pad, stream, 3);
}

// ---- half-tensor forms (precision-16 training step with its activations AND a per-step copy of the weights stored as halves, as
// torch.autocast keeps them under pl.Trainer(precision=16), tools.py:263): fp16 operands straight from memory, fp32 accumulation,
// the output rounded once.  Same call sites as ssad_conv_igemm_fwd_stats / ssad_conv_igemm_dgrad. ----
extern "C" int ssad_conv_igemm_fwd_stats_h(const void* in, const void* w_ohwi, void* out, int64_t N, int H, int W, int Cin,
                                           int Cout, int KH, int KW, int stride, int pad, float eps, float momentum,
                                           float* mean, float* invstd, float* running_mean, float* running_var,
                                           double* workspace, void* stream) {
    SSAD_CHECK_ARG(mean && invstd && workspace, "null pointer");
    SSAD_CHECK_ARG((((uintptr_t)in | (uintptr_t)w_ohwi | (uintptr_t)out) & 15) == 0, "half tensors and filters must be 16-byte aligned");
    int rows = 0;
    int rc = conv_fwd_impl((const float*)in, (const float*)w_ohwi, (float*)out, nullptr, nullptr, nullptr, 0, N, H, W, Cin, Cout, KH, KW,
                           stride, pad, 0, stream, 2, workspace, &rows, 1);
    if (rc) return rc;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    return ssad_bn_finalize_partials(workspace, rows, N * Ho * Wo, Cout, eps, momentum, mean, invstd, running_mean,
                                     running_var, stream);
}

extern "C" int ssad_conv_igemm_dgrad_h(const void* dy, const void* w_flipT, void* dx, const void* residual, int64_t N,
                                       int Hy, int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride,
                                       int pad, void* stream) {
    SSAD_CHECK_ARG((((uintptr_t)dy | (uintptr_t)w_flipT | (uintptr_t)dx | (uintptr_t)residual) & 15) == 0,
                   "half tensors and filters must be 16-byte aligned");
    return dgrad_impl((const float*)dy, (const float*)w_flipT, (float*)dx, (const float*)residual, N, Hy, Wy, Cout, Hx, Wx, Cin, KH, KW,
                      stride, pad, stream, 2, nullptr, 1);
}
