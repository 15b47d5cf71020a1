// 3x3 / stride 1 / pad 1 convolution over HALF tensors (NHWC fp16 in HBM, OHWI fp16 weights, fp32 accumulation) for the
// precision-16 training step: the torchvision BasicBlock conv3x3 layers (models.py:224 of the reference under
// pl.Trainer(precision=16), tools.py:263) forward, and -- with the flipped filter -- their input gradients.
//
// With 16-bit operands the matrix work of a ResNet-18 step is ~0.2 ms; what the implicit GEMM (conv_igemm.hip, IO = true) spends
// is staging: every activation row once per filter tap, one barrier per 32-deep K-step, an LDS-transpose epilogue (104 us per
// layer2-4 conv at batch 256, 220 us per layer1 conv on the c64 kernel whose weight slices wait for the L2 once per tap).  Here:
//   * a workgroup owns 128 output pixels (8 x 16 of one image; two 8 x 8 maps on the 8 x 8 maps of layer4) x BN output channels and
//     walks the input channels in chunks of 64: the (8+2) x (16+2) x 64 halo of a chunk is staged ONCE for its nine taps -- a tap is
//     an LDS address immediate -- optionally through relu(bn(x)) of the producing layer (train-mode BatchNorm + ReLU applied on
//     load, in fp32, rounded once: the normalised activation never exists in HBM);
//   * the [BN][64] weight slice of a (tap, chunk) step is the only thing staged per step: 16-byte pieces, requested TWO steps ahead
//     into two register sets (an L2 round trip is longer than one step's MFMAs), double-buffered in LDS, one barrier per step of
//     16 MFMAs per wave;
//   * workgroups are persistent (at most two per CU) over (tile, chunk) pairs: the next pair's halo is in flight in registers
//     during the nine steps of the current one, so a tile never waits for HBM;
//   * the epilogue stores accumulators straight from registers, every lane its own halves (a wave store covers whole 64-byte
//     runs: 32 consecutive channels of one pixel per lane half); ragged tiles pair neighbouring lanes (one DPP move) and predicate
//     per pixel.  BatchNorm statistics are those of the STORED halves, summed per lane in double across all tiles of the
//     workgroup: one partial row per workgroup (<= 512 rows for the finalize kernel instead of one per tile).
//
// Where its time goes (round 5, tools/micro/conv16_ablate.hip, 256 x 32 x 32 x 128 -> 128, zero operands; 31 us = the fp16 MFMA floor):
// 109-118 us as shipped; without the MFMAs 47-50; MFMAs + fragment reads + barriers alone 49 -- the phases ADD, the co-resident
// workgroup does not fill the matrix stream's gaps (the finding of profiles/r03_igemm_phases.md for fp32 holds for 16-bit MFMAs);
// without the weight-slice loads 70 (their L2 latency: three steps of prefetch and a per-workgroup tap rotation took 132 -> 81-109);
// without the epilogue 71, and the epilogue WITHOUT ITS STORES costs the same as with them: it is ~1 500 VALU instructions per tile
// (64-bit offsets, lane-pair selects, conversions for the statistics).  A version with 32-bit scalar offsets and statistics taken
// from the packed halves was built: the register allocator then spills 30-100 VGPRs (64 accumulators + 48 weight + 24 halo
// prefetch registers are live across the epilogue) and nothing is gained (115 us); dealing the weight pieces out between the MFMA
// groups: 117 us.  Not kept.  Kept: whole tiles store single halves per lane instead of exchanged pairs (no DPP move, no selects;
// the same 64-byte runs): the 26 launches of a batch-256 step 3.29 -> 3.15 ms, same box, A / B.
#include "common.h"
#include <stdlib.h>

// Ablation switches for tools/micro/conv16_ablate.hip (timing only, results are garbage); always 0 in the library build.
//   1 = no weight-slice loads   2 = no halo loads   4 = no epilogue   8 = no MFMAs   16 = no fragment reads   32 = no step barriers
//   64 = the epilogue without its global stores
#ifndef CONV16_ABL
#define CONV16_ABL 0
#endif

namespace {

constexpr int CK = 64;                 // input channels per chunk
constexpr int LDP = CK + 8;            // halves per LDS row (144 B: conflict-free 16-byte fragment reads)

struct Conv16Params {
    const hf* in;            // [N][H][W][Cin]
    const hf* wt;            // [Cout][3][3][Cin]
    hf* out;                 // [N][H][W][Cout]
    const hf* residual;      // optional [N][H][W][Cout], added before the rounding
    const float* tr_mean;    // optional input transform x <- relu((x - mean) * invstd * gamma + beta), per input channel
    const float* tr_invstd;
    const float* tr_gamma;
    const float* tr_beta;
    hf* emit;                // optional (with a transform): the transformed input [N][H][W][Cin], written once (channel slab 0)
    double* stats;           // optional [gridDim.x][2][Cout]
    int N, H, W, Cin, Cout;
    int tiles_y, tiles_x, nchunks;
    int64_t ntiles;          // TW16: N * tiles_y * tiles_x;  TW8: ceil(N * tiles_y / 2) (two 8 x 8 sub-tiles per tile)
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// value of the neighbouring lane (lane ^ 1) as one DPP register move (quad_perm [1, 0, 3, 2]).  Inline assembly on purpose: hipcc's DPP
// combiner folded two __builtin_amdgcn_update_dpp calls of an (even ? a : b) pattern into one move of the WRONG source register
// (round 5, found by tests/test_hip_half.py::test_conv3x3_h_halo_kernel); s_nop 1 covers the VALU-write -> DPP-read hazard.
__device__ __forceinline__ float dpp_swap_pair(float x) {
    float r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    return r;
}

// BN: output channels per workgroup (64 or 128).  TW8: maps at most 8 wide -- a tile is two 8 x 8 sub-tiles (consecutive
// (image, tile row) pairs), each with its own 10 x 10 halo; otherwise one 8 x 16 block with an 10 x 18 halo.
template <int BN, bool TW8>
__global__ __launch_bounds__(256, 2) void conv3x3_h_kernel(Conv16Params p) {
    constexpr int TN = BN / 64;                        // 32-wide accumulator tiles per wave along the channels (waves: 2 x 2)
    constexpr int HW_ = TW8 ? 10 : 18;                 // halo row length in pixels
    constexpr int NHP = TW8 ? 200 : 180;               // halo pixels
    constexpr int NHI = (NHP + 31) / 32;               // halo pixels per thread (8 threads per pixel: one 16-byte piece each)
    constexpr int NBI = BN / 32;                       // weight rows per thread per step
    constexpr int HALO_H = NHP * LDP;                  // halves
    constexpr int BS_H = BN * LDP;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    hf* halo = (hf*)lds;
    hf* Bs = halo + HALO_H;                            // [2][BN][LDP]
    float* trp = (float*)(Bs + 2 * BS_H);              // [4][Cin]: mean, invstd, gamma, beta of the input transform (when given)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int co0 = blockIdx.y * BN;
    const int piece = tid & 7, prow = tid >> 3;        // staging role: 16-byte piece of a 64-half row, rows prow + 32 i

    // ---- tiles of this workgroup: t = blockIdx.x, blockIdx.x + gridDim.x, ... ----
    const int64_t my_tiles = p.ntiles > (int64_t)blockIdx.x ? (p.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int64_t nfill = my_tiles * p.nchunks;
    const int tpi = p.tiles_y * p.tiles_x;

    // position of a sub-tile's first output pixel: TW16: tile -> (n, y0, x0); TW8: sub-tile st = 2 tile + j -> (n, y0), x0 = 0
    auto tile_origin = [&](int64_t tile, int j, int& n, int& y0, int& x0) -> bool {
        if (TW8) {
            const int64_t st = 2 * tile + j;
            n = (int)(st / p.tiles_y);
            y0 = (int)(st - (int64_t)n * p.tiles_y) * 8;
            x0 = 0;
            return n < p.N;
        }
        n = (int)(tile / tpi);
        const int rem = (int)(tile - (int64_t)n * tpi);
        y0 = (rem / p.tiles_x) * 8;
        x0 = (rem % p.tiles_x) * 16;
        return true;
    };

    // ---- halo prefetch registers ----
    u32x4 hreg[NHI];
    bool hin[NHI];
    auto load_halo = [&](int64_t tile_index, int chunk) {           // tile_index-th tile of this workgroup
        const int64_t tile = (int64_t)blockIdx.x + tile_index * gridDim.x;
        int n0, ya, xa, n1 = 0, yb = 0, xb = 0;
        bool ok1 = false;
        const bool ok0 = tile_origin(tile, 0, n0, ya, xa);
        if (TW8) ok1 = tile_origin(tile, 1, n1, yb, xb);
#pragma unroll
        for (int i = 0; i < NHI; ++i) {
            const int hp = prow + 32 * i;
            int n, y, x;
            bool ok = hp < NHP;
            if (TW8) {
                const int j = hp >= 100, q = hp - 100 * j;
                const int hy = q / 10, hx = q - 10 * hy;
                n = j ? n1 : n0;
                y = (j ? yb : ya) - 1 + hy;
                x = -1 + hx;
                ok = ok && (j ? ok1 : ok0);
            } else {
                const int hy = hp / 18, hx = hp - 18 * hy;
                n = n0;
                y = ya - 1 + hy;
                x = xa - 1 + hx;
                ok = ok && ok0;
            }
            ok = ok && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            hin[i] = ok;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok && !(CONV16_ABL & 2)) v = *(const u32x4*)(p.in + (((int64_t)n * p.H + y) * p.W + x) * p.Cin + chunk * CK + piece * 8);
            hreg[i] = v;
        }
    };
    auto store_halo = [&](int64_t tile_index, int chunk) {
        if (p.tr_mean) {
            // producer's train-mode BatchNorm + ReLU on load: bn_apply_fwd's expression in fp32, rounded once.  Zero padding pads the
            // TRANSFORMED activation: out-of-image pieces stay zero.
            const int c = chunk * CK + piece * 8;
            float mu[8], sc[8], ga[8], be[8];
#pragma unroll
            for (int k = 0; k < 8; k += 4) {
                const f32x4 a = *(const f32x4*)(trp + c + k), b = *(const f32x4*)(trp + p.Cin + c + k);
                const f32x4 g = *(const f32x4*)(trp + 2 * p.Cin + c + k), e = *(const f32x4*)(trp + 3 * p.Cin + c + k);
#pragma unroll
                for (int q = 0; q < 4; ++q) { mu[k + q] = a[q]; sc[k + q] = b[q]; ga[k + q] = g[q]; be[k + q] = e[q]; }
            }
#pragma unroll
            for (int i = 0; i < NHI; ++i) {
                if (!hin[i]) continue;
                f16x8 v = __builtin_bit_cast(f16x8, hreg[i]);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (hf)fmaxf(((float)v[k] - mu[k]) * sc[k] * ga[k] + be[k], 0.f);
                hreg[i] = __builtin_bit_cast(u32x4, v);
            }
            if (p.emit && blockIdx.y == 0) {
                // the activation the weight gradient of this layer reads: interior pixels of the halo, once per (tile, chunk)
                const int64_t tile = (int64_t)blockIdx.x + tile_index * gridDim.x;
                int n0, ya, xa, n1 = 0, yb = 0, xb = 0;
                tile_origin(tile, 0, n0, ya, xa);
                if (TW8) tile_origin(tile, 1, n1, yb, xb);
#pragma unroll
                for (int i = 0; i < NHI; ++i) {
                    const int hp = prow + 32 * i;
                    if (!hin[i]) continue;
                    int n, y, x, hy, hx;
                    if (TW8) {
                        const int j = hp >= 100, q = hp - 100 * j;
                        hy = q / 10; hx = q - 10 * hy;
                        n = j ? n1 : n0; y = (j ? yb : ya) - 1 + hy; x = -1 + hx;
                    } else {
                        hy = hp / 18; hx = hp - 18 * hy;
                        n = n0; y = ya - 1 + hy; x = xa - 1 + hx;
                    }
                    if (hy >= 1 && hy <= 8 && hx >= 1 && hx <= (TW8 ? 8 : 16))
                        *(u32x4*)(p.emit + (((int64_t)n * p.H + y) * p.W + x) * p.Cin + c + 0) = hreg[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NHI; ++i) {
            const int hp = prow + 32 * i;
            if (hp < NHP) *(u32x4*)(halo + hp * LDP + piece * 8) = hreg[i];
        }
    };

    // ---- weight slices: step s = (fill, tap) -> rows co0 + prow + 32 i, 64 halves at [tap][chunk * 64] ----
    u32x4 breg[3][NBI];         // three register sets: a slice is requested THREE steps before it is consumed, stored to LDS one step before
    const hf* wrow = p.wt + ((int64_t)(co0 + prow) * 9) * p.Cin + piece * 8;
    auto load_b = [&](int tap, int chunk, int set) {
        const hf* src = wrow + (int64_t)tap * p.Cin + chunk * CK;
        if (CONV16_ABL & 1) return;
#pragma unroll
        for (int i = 0; i < NBI; ++i) breg[set][i] = *(const u32x4*)(src + (int64_t)i * 32 * 9 * p.Cin);
    };
    auto store_b = [&](int buf, int set) {
        hf* dst = Bs + buf * BS_H + prow * LDP + piece * 8;
#pragma unroll
        for (int i = 0; i < NBI; ++i) *(u32x4*)(dst + i * 32 * LDP) = breg[set][i];
    };

    // ---- fragment bases ----
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pp = wm * 64 + i * 32 + r;
        if (TW8) abase[i] = (((pp >> 6) * 100) + ((pp & 63) >> 3) * 10 + (pp & 7)) * LDP + 8 * h;
        else abase[i] = ((pp >> 4) * 18 + (pp & 15)) * LDP + 8 * h;
    }
    const int bbase = (wn * 32 * TN + r) * LDP + 8 * h;

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    double st0[TN], st1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) st0[j] = st1[j] = 0.0;

    if (p.tr_mean) {
        for (int c = tid; c < p.Cin; c += 256) {
            trp[c] = p.tr_mean[c]; trp[p.Cin + c] = p.tr_invstd[c]; trp[2 * p.Cin + c] = p.tr_gamma[c]; trp[3 * p.Cin + c] = p.tr_beta[c];
        }
        __syncthreads();
    }
    const int64_t nstep = nfill * 9;
    // Every workgroup walks the nine taps of a chunk in its own rotation: the ~512 persistent workgroups run in lockstep (same work,
    // same start), and with one common order all of them ask the L2 for the same 16 KB weight slice at the same moment -- the
    // slice's lines are then served one requester after the other.  With nine rotations a slice has a ninth of the requesters.
    // (fp32 accumulation order differs between workgroups by the rotation only: deterministic for a given launch geometry.)
#ifdef CONV16_NO_ROT
    const int rot = 0;
#else
    const int rot = (int)((blockIdx.x + blockIdx.y) % 9);
#endif
    auto rtap = [&](int t) { const int u = t + rot; return u >= 9 ? u - 9 : u; };
    // (rotating the channel chunks as well was measured: no gain on layers 2-3, and the 8 x 8 x 512 layer, whose 4.7 MB of weights
    // do not fit an XCD's L2, lost a third -- more distinct slices in flight)
    auto rchunk = [&](int c) { return c; };
    if (nfill > 0) {
        load_halo(0, rchunk(0));
        load_b(rtap(0), rchunk(0), 0);
        store_b(0, 0);
        load_b(rtap(1), rchunk(0), 1);
        load_b(rtap(2), rchunk(0), 2);
    }
    int chunk_cur = 0;
    int64_t tile_i = 0;

    for (int64_t f = 0; f < nfill; ++f) {
        // every wave is done with the previous halo (the last step of the previous fill ended on a barrier)
        store_halo(tile_i, rchunk(chunk_cur));
        if (f + 1 < nfill) {
            if (chunk_cur + 1 < p.nchunks) load_halo(tile_i, rchunk(chunk_cur + 1));
            else load_halo(tile_i + 1, rchunk(0));
        }
        __syncthreads();                       // halo(f) and the weight slice of its first step are visible
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int64_t s = f * 9 + tap;
            const int cur = (int)(s & 1);
            // request the slice three steps ahead into the register set that was stored during the previous step (s % 3 == tap % 3:
            // a fill has nine steps)
            if (s + 3 < nstep) {
                if (tap + 3 < 9) load_b(rtap(tap + 3), rchunk(chunk_cur), tap % 3);
                else load_b(rtap(tap + 3 - 9), rchunk(chunk_cur + 1 == p.nchunks ? 0 : chunk_cur + 1), tap % 3);
            }
            const int te = rtap(tap);
            const int ty3 = te >= 6 ? 2 : te >= 3 ? 1 : 0;
            const int toff = (ty3 * HW_ + (te - 3 * ty3)) * LDP;
            const hf* Ab = halo + toff;
            const hf* Bb = Bs + cur * BS_H + bbase;
#pragma unroll
            for (int kk = 0; kk < CK / 16; ++kk) {
                f16x8 a[2], b[TN];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = *(const f16x8*)(Ab + abase[i] + ((CONV16_ABL & 16) ? 0 : kk * 16));
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *(const f16x8*)(Bb + j * 32 * LDP + ((CONV16_ABL & 16) ? 0 : kk * 16));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (!(CONV16_ABL & 8)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            // the slice of the next step (requested two steps ago) goes to the other LDS stage: nobody reads it before the barrier
            if (s + 1 < nstep) store_b(cur ^ 1, (tap + 1) % 3);
            if (!(CONV16_ABL & 32)) __syncthreads();
        }
        if (++chunk_cur < p.nchunks) continue;
        chunk_cur = 0;

        // ---- epilogue of a finished tile: straight from the accumulators ----
        const int64_t tile = (int64_t)blockIdx.x + tile_i * gridDim.x;
        ++tile_i;
        int n0, ya, xa, n1 = 0, yb = 0, xb = 0;
        bool ok1 = false;
        const bool ok0 = tile_origin(tile, 0, n0, ya, xa);
        if (TW8) ok1 = tile_origin(tile, 1, n1, yb, xb);
        if (CONV16_ABL & 4) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { sum += acc[i][j][e]; acc[i][j][e] = 0.f; }
            if (sum == 123.456f) p.out[0] = (hf)sum;
            continue;
        }
        const bool even = (r & 1) == 0;
        // whole tiles (every pixel inside its map: all tiles of the ResNet-18 maps) take a path without a single per-pixel predicate
        const bool full = TW8 ? (p.W == 8 && ok0 && ok1 && ya + 8 <= p.H && yb + 8 <= p.H) : (ya + 8 <= p.H && xa + 16 <= p.W);
        if (full) {
            // Every lane stores its own 16 x TN values of a tile as single halves: a wave's store still covers whole 64-byte runs
            // (32 lanes = 32 consecutive channels of one pixel, the lane halves two pixels), exactly what the lane-pair exchange of
            // the ragged path produces -- without its DPP move and three selects per pair.  Measured reason: this epilogue's VALU
            // instructions, not its stores, were a third of the kernel (conv16_ablate, bits 4 / 64).
            const int64_t rowst = (int64_t)p.W * p.Cout;
            int64_t base;                           // this lane's pixel for register 0 of tile i = 0, its channel of tile j = 0
            if (TW8) base = (((int64_t)(wm ? n1 : n0) * p.H + (wm ? yb : ya)) * p.W + 4 * h) * p.Cout;
            else base = (((int64_t)n0 * p.H + ya + wm * 4) * p.W + xa + 4 * h) * p.Cout;
            base += co0 + wn * 32 * TN + r;
            hf* const outp = p.out + base;
            const hf* const resp = p.residual ? p.residual + base : nullptr;
            float fs[TN], fq[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) fs[j] = fq[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    // pixel of register e = 8 g + q (lane half h = 0): TW16: row 2 i + g, column 8 (q >> 2) + (q & 3); TW8: row
                    // 4 i + 2 g + (q >> 2), column q & 3 -- wave-uniform offsets; eight registers at a time (register pressure)
                    int64_t o[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        o[q] = TW8 ? (int64_t)(4 * i + 2 * g + (q >> 2)) * rowst + (int64_t)(q & 3) * p.Cout
                                   : (int64_t)(2 * i + g) * rowst + (int64_t)(8 * (q >> 2) + (q & 3)) * p.Cout;
                    hf rres[8][TN];
                    if (p.residual) {               // the residual loads of these eight pixels in flight before the first is used
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int j = 0; j < TN; ++j) rres[q][j] = resp[o[q] + 32 * j];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            float v = acc[i][j][8 * g + q];
                            if (p.residual) v += (float)rres[q][j];
                            const hf ov = (hf)v;
                            if (!(CONV16_ABL & 64) || v == 123.456f)       // ablation 64: the epilogue's arithmetic without its stores
                                outp[o[q] + 32 * j] = ov;
                            if (p.stats) {          // statistics of what is stored
                                const float sv = (float)ov;
                                fs[j] += sv;
                                fq[j] += sv * sv;
                            }
                        }
                }
            if (p.stats) {
#pragma unroll
                for (int j = 0; j < TN; ++j) { st0[j] += (double)fs[j]; st1[j] += (double)fq[j]; }
            }
        } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                // registers e, e + 1 are rows m, m + 1 of this lane's column; the lane pair (r, r ^ 1) swaps one value so that the even
                // lane owns row m and the odd lane row m + 1, two neighbouring channels each
                const int m = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h + (even ? 0 : 1);
                int n, y, x;
                bool ok;
                if (TW8) {
                    const int j = m >> 6, q = m & 63;
                    n = j ? n1 : n0;
                    y = (j ? yb : ya) + (q >> 3);
                    x = q & 7;
                    ok = j ? ok1 : ok0;
                } else {
                    n = n0;
                    y = ya + (m >> 4);
                    x = xa + (m & 15);
                    ok = ok0;
                }
                ok = ok && y < p.H && x < p.W;
                const int64_t pix = ((int64_t)n * p.H + y) * p.W + x;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float mine = even ? acc[i][j][e] : acc[i][j][e + 1];
                    const float give = even ? acc[i][j][e + 1] : acc[i][j][e];
                    const float got = __shfl_xor(give, 1);
                    float v0 = even ? mine : got, v1 = even ? got : mine;          // channels c, c + 1 with c = the pair's even column
                    const int c = co0 + wn * 32 * TN + j * 32 + (r & ~1);
                    if (ok) {
                        const int64_t o = pix * p.Cout + c;
                        if (p.residual) {
                            const unsigned rr = *(const unsigned*)(p.residual + o);
                            typedef hf h2 __attribute__((ext_vector_type(2)));
                            const h2 rv = __builtin_bit_cast(h2, rr);
                            v0 += (float)rv[0];
                            v1 += (float)rv[1];
                        }
                        typedef hf h2 __attribute__((ext_vector_type(2)));
                        const h2 ov = {(hf)v0, (hf)v1};
                        *(unsigned*)(p.out + o) = __builtin_bit_cast(unsigned, ov);
                    }
                    if (p.stats) {
                        // statistics per CHANNEL need every row of the channel: a lane sums its own column over both rows of the pair
                        // (its own register of the other row is rounded here exactly as the partner rounds it for the store)
                        const int mo = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;         // rows of registers e, e + 1
                        float a0 = (float)(hf)acc[i][j][e], a1 = (float)(hf)acc[i][j][e + 1];
                        bool k0, k1;
                        if (TW8) {
                            const int j0 = mo >> 6, q0 = mo & 63, j1 = (mo + 1) >> 6, q1 = (mo + 1) & 63;
                            k0 = (j0 ? ok1 : ok0) && (j0 ? yb : ya) + (q0 >> 3) < p.H && (q0 & 7) < p.W;
                            k1 = (j1 ? ok1 : ok0) && (j1 ? yb : ya) + (q1 >> 3) < p.H && (q1 & 7) < p.W;
                        } else {
                            k0 = ok0 && ya + (mo >> 4) < p.H && xa + (mo & 15) < p.W;
                            k1 = ok0 && ya + ((mo + 1) >> 4) < p.H && xa + ((mo + 1) & 15) < p.W;
                        }
                        a0 = k0 ? a0 : 0.f;
                        a1 = k1 ? a1 : 0.f;
                        st0[j] += (double)(a0 + a1);
                        st1[j] += (double)(a0 * a0 + a1 * a1);
                    }
                }
            }
        }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }

    if (p.stats) {
        // lane halves, then the two row-waves in a fixed order through LDS (every stage is dead: the loop ended on a barrier)
        double* S = (double*)lds;                  // [2 wm][2][BN]
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
        for (int u = tid; u < 2 * BN; u += 256) {
            const int which = u / BN, cc = u % BN;
            p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + co0 + cc] = S[(0 * 2 + which) * BN + cc] + S[(1 * 2 + which) * BN + cc];
        }
    }
}

struct Geo {
    bool tw8;
    int tiles_y, tiles_x, bn, gx, gy;
    int64_t ntiles;
};

static Geo geometry(int64_t N, int H, int W, int Cout) {
    Geo g;
    g.tw8 = W <= 8;
    g.tiles_y = (H + 7) / 8;
    g.tiles_x = g.tw8 ? 1 : (W + 15) / 16;
    g.ntiles = g.tw8 ? (N * g.tiles_y + 1) / 2 : N * g.tiles_y * g.tiles_x;
    g.bn = Cout % 128 == 0 ? 128 : 64;
    g.gy = Cout / g.bn;
    // persistent workgroups: two per CU over all channel slabs
    static const int slots = getenv("SSAD_CONV16_WGS") ? atoi(getenv("SSAD_CONV16_WGS")) : 512;
    int64_t gx = slots / g.gy;
    if (gx < 1) gx = 1;
    if (gx > g.ntiles) gx = g.ntiles;
    g.gx = (int)gx;
    return g;
}

}  // namespace

// 1 when ssad_conv3x3_h handles the layer (channel counts multiples of 64).
extern "C" int ssad_conv3x3_h_ok(int Cin, int Cout) {
    static const int on = getenv("SSAD_CONV16") ? atoi(getenv("SSAD_CONV16")) : 1;
    return on && Cin % 64 == 0 && Cout % 64 == 0;
}

// rows of the statistics workspace (x 2 x Cout doubles)
extern "C" int64_t ssad_conv3x3_h_stats_rows(int64_t N, int H, int W, int Cout) {
    return geometry(N, H, W, Cout).gx;
}

// out = conv3x3(pad 1, stride 1)(T(in)) (+ residual), half tensors NHWC, half weights OHWI [Cout][3][3][Cin]; T = identity or
// relu((x - tr_mean) * tr_invstd * tr_gamma + tr_beta) per input channel (the producer's train-mode BatchNorm + ReLU, applied on load).
// emit (optional, with a transform): receives T(in), the activation this layer's weight gradient reads.
// stats_ws != NULL: train-mode BatchNorm statistics of the stored output (ssad_conv3x3_h_stats_rows(...) * 2 * Cout doubles),
// finalised as ssad_conv_igemm_fwd_stats does.  As the input gradient: in = dz, w = the flipped filter [Cin][3][3][Cout] with the
// roles of Cin / Cout swapped, residual = the identity-branch gradient.
extern "C" int ssad_conv3x3_h(const void* in, const void* w_ohwi, void* out, const void* residual, const float* tr_mean,
                              const float* tr_invstd, const float* tr_gamma, const float* tr_beta, void* emit, int64_t N, int H, int W, int Cin,
                              int Cout, double* stats_ws, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                              float* running_var, void* stream) {
    SSAD_CHECK_ARG(in && w_ohwi && out && N > 0 && H > 0 && W > 0, "bad argument");
    SSAD_CHECK_ARG(Cin % 64 == 0 && Cout % 64 == 0, "channel counts must be multiples of 64");
    SSAD_CHECK_ARG((((uintptr_t)in | (uintptr_t)w_ohwi | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)emit) & 15) == 0,
                   "half tensors and filters must be 16-byte aligned (they are read in 16-byte pieces)");
    SSAD_CHECK_ARG(!tr_mean || (tr_invstd && tr_gamma && tr_beta), "input transform needs mean, invstd, gamma, beta");
    SSAD_CHECK_ARG(!stats_ws || (mean && invstd), "statistics need mean / invstd outputs");
    SSAD_CHECK_ARG(!emit || tr_mean, "emit without an input transform");
    SSAD_CHECK_ARG(N * (int64_t)H * W < (int64_t)1 << 31, "too many pixels for one launch");
    const Geo g = geometry(N, H, W, Cout);
    Conv16Params p;
    p.in = (const hf*)in; p.wt = (const hf*)w_ohwi; p.out = (hf*)out; p.residual = (const hf*)residual;
    p.tr_mean = tr_mean; p.tr_invstd = tr_invstd; p.tr_gamma = tr_gamma; p.tr_beta = tr_beta; p.emit = (hf*)emit;
    p.stats = stats_ws;
    p.N = (int)N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.tiles_y = g.tiles_y; p.tiles_x = g.tiles_x; p.nchunks = Cin / CK; p.ntiles = g.ntiles;
    const dim3 grid((unsigned)g.gx, (unsigned)g.gy);
    hipStream_t st = (hipStream_t)stream;
    const int lds_bytes = ((g.tw8 ? 200 : 180) * LDP + 2 * g.bn * LDP) * 2 + (tr_mean ? 16 * Cin : 0);
    SSAD_CHECK_ARG(Cin <= 1024, "at most 1024 input channels (transform table in LDS)");
    static bool attr_set = false;
    if (!attr_set) {
        SSAD_SET_DYN_LDS((conv3x3_h_kernel<128, true>), (200 * LDP + 2 * 128 * LDP) * 2 + 16 * 1024);
        SSAD_SET_DYN_LDS((conv3x3_h_kernel<128, false>), (180 * LDP + 2 * 128 * LDP) * 2 + 16 * 1024);
        SSAD_SET_DYN_LDS((conv3x3_h_kernel<64, true>), (200 * LDP + 2 * 64 * LDP) * 2 + 16 * 1024);
        SSAD_SET_DYN_LDS((conv3x3_h_kernel<64, false>), (180 * LDP + 2 * 64 * LDP) * 2 + 16 * 1024);
        attr_set = true;
    }
    if (g.bn == 128) {
        if (g.tw8) hipLaunchKernelGGL((conv3x3_h_kernel<128, true>), grid, dim3(256), lds_bytes, st, p);
        else hipLaunchKernelGGL((conv3x3_h_kernel<128, false>), grid, dim3(256), lds_bytes, st, p);
    } else {
        if (g.tw8) hipLaunchKernelGGL((conv3x3_h_kernel<64, true>), grid, dim3(256), lds_bytes, st, p);
        else hipLaunchKernelGGL((conv3x3_h_kernel<64, false>), grid, dim3(256), lds_bytes, st, p);
    }
    SSAD_CHECK_LAUNCH();
    if (stats_ws)
        return ssad_bn_finalize_partials(stats_ws, g.gx, N * H * W, Cout, eps, momentum, mean, invstd, running_mean, running_var, stream);
    return 0;
}
