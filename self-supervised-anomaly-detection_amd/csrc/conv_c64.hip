// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (ResNet-18 layer1: four forward convs and
// four input-gradient convs per training step at 64x64 maps, one third of the step's FLOPs) on the fp32 matrix cores,
// as a HALO-TILE direct convolution.
//
// The implicit-GEMM kernel (conv_igemm.hip) re-stages every activation row once per filter tap: per 32 MFMAs a thread
// issues 5 global loads + 5 LDS stores (Cout = 64 tile), and the wave's in-order stream cannot hide them: 85-100 TFLOP/s
// on this shape against 125-132 on the wide layers.  Here a workgroup owns an 8 x 16 patch of output pixels of ONE image
// and all 64 output channels:
//   * the 10 x 18 x 64 input halo is loaded ONCE into LDS (51 200 B: 68-float pixels, 1 280-float rows, + the 17 408 B weight slice
//     of a tap = 68 608 B per workgroup; two workgroups per CU = 134 KB, which only gfx950's 160 KB of LDS per CU holds; zero padding
//     written as zeros, so there are no tap masks), optionally through a per-channel affine + ReLU (the train-mode BatchNorm + ReLU of the producing layer:
//     the normalised activation never makes a round trip through HBM) and optionally emitted for the weight-gradient
//     kernel that needs it later;
//   * the A fragment of tap (ky, kx) is the same LDS image read at a constant byte offset (an instruction immediate);
//   * only the 64 x 64 weight slice of a tap (16 KB) is staged per step: 4 loads + 4 LDS stores per 64 MFMAs.
// Train-mode BatchNorm statistics of the output are taken from the accumulators in registers (fp64, fixed order).
//
// Replaces the same autograd nodes as conv_igemm.hip for this shape: torchvision BasicBlock conv3x3 (models.py:224 of
// the reference, under trainer.fit) and its input gradient.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

// Ablation switches for tools/micro/c64_ablate.hip (where the time goes); always 0 in the library build.
#ifndef C64_ABL
#define C64_ABL 0
#endif
// 1 = tiles inside the image store their accumulators directly (no LDS transpose); 0 = always the LDS epilogue
#ifndef C64_DIRECT_EPI
#define C64_DIRECT_EPI 1
#endif

#if C64_ABL & 16
__device__ unsigned long long* g_c64_trace;      // [workgroups][8]: phase time stamps (tools/micro/c64_ablate.hip)
#define C64_STAMP(i) do { if (threadIdx.x == 0 && g_c64_trace) g_c64_trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define C64_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int TH = 8, TW = 16;              // output pixels per workgroup
constexpr int HH = TH + 2, HW = TW + 2;     // halo
constexpr int C = 64;                       // channels in and out
constexpr int LDP = C + 4;                  // LDS row pitch in floats (272 B: conflict-free ds_read_b128 over rows)
// halo row pitch: HW pixels + 56 floats of padding, so that the pitch is a multiple of 64 dwords (18 * 68 + 56 = 1280): the two tile
// rows a wave reads in one ds_read_b128 (lanes 0-15 / 16-31) then fall on the same bank pattern shifted by whole pixels and the
// 16-lane groups of the instruction stay conflict-free (round 2: pitch 1224 = 8 mod 64, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.19)
constexpr int ROWP = HW * LDP + 56;
constexpr int HALO_F = HH * ROWP;           // floats
constexpr int WT_F = C * LDP;               // one tap's [co][ci] slice
constexpr int LDS_BYTES = (HALO_F + WT_F) * 4;

struct C64Params {
    const float* in;          // [N][H][W][64]
    const float* wt;          // [64][3][3][64]  (OHWI)
    float* out;               // [N][H][W][64]
    const float* residual;    // optional, added to the output ...
    const uint8_t* res_mask;  // ... under this nibble mask when given (one byte per channel quad, bit k = keep channel 4q+k)
    const float* tr_mean;     // optional input transform: x <- relu((x - mean) * invstd * gamma + beta)
    const float* tr_invstd;
    const float* tr_gamma;
    const float* tr_beta;
    float* emit;              // optional: the transformed input, written for the interior pixels
    double* stats;            // optional [workgroups][2][64] column sums / sums of squares of the raw output
    int N, H, W, tiles_y, tiles_x;
    // EVAL instantiation (frozen-BatchNorm inference, the patch-scoring trunk): out = act(conv * scale + shift + residual),
    // tensors addressed through (pixel, sample) strides in floats so that NHWC and the position-major [H][W][N][C] both fit
    const float* scale;
    const float* shift;
    int relu;
    int64_t in_ps, in_ss, out_ps, out_ss, res_ps, res_ss;
    int stagger;              // start delay of the workgroups in the second wave slot of their CU, in units of s_sleep(127)
};

// OP: 0 = exact fp32 (v_mfma_f32_32x32x2_f32); 1 / 2 = bf16 / fp16 OPERANDS, fp32 accumulation (v_mfma_f32_32x32x16_*): the layer1
// convolutions of the precision-16 step (pl.Trainer(precision=16), tools.py:263).  Tensors stay fp32 in memory; the halo and the
// weight slice are rounded while they are staged (the input transform and `emit` run in fp32 before the rounding), pixels and filter
// rows are 72 halves apart in LDS (144 B: conflict-free b128 fragment reads), 8 MFMAs per tap instead of 128.  Accumulators have the
// fp32 kernel's layout, so every epilogue below is shared.
template <int OP> struct C64Op { using t = float; using v4 = f32x4; using v8 = f32x4; };
template <> struct C64Op<1> {
    using t = __bf16; using v4 = bf16x4; using v8 = bf16x8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct C64Op<2> {
    using t = _Float16; using v4 = f16x4; using v8 = f16x8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
constexpr int LDP16 = C + 8;                        // halves per pixel / filter row of the 16-bit forms
constexpr int ROWP16 = HW * LDP16 + 8;              // halves per halo row
constexpr int HALO16_H = HH * ROWP16, WT16_H = C * LDP16;
constexpr int LDS_BYTES16 = ((HALO16_H + WT16_H) * 2 > 128 * LDP * 4 ? (HALO16_H + WT16_H) * 2 : 128 * LDP * 4);   // >= the LDS epilogue tile

// TI = hf (OP == 2 only): in, out, residual and emit are stored as halves -- the precision-16 step with its tensors as autocast keeps
// them; 8-byte accesses, the output rounded once, statistics of the stored halves, always the LDS epilogue (8-byte row pieces).
template <bool EVAL, int OP = 0, typename TI = float>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(C64Params p) {
    static_assert(!(EVAL && OP), "the 16-bit forms are training-only");
    static_assert(std::is_same<TI, float>::value || OP == 2, "half tensors go with fp16 operands");
    constexpr bool HIO = !std::is_same<TI, float>::value;
    using op_t = typename C64Op<OP>::t;
    using op4 = typename C64Op<OP>::v4;
    using op8 = typename C64Op<OP>::v8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* halo = lds;
    float* Bs = lds + HALO_F;
    op_t* halo16 = (op_t*)lds;                      // OP != 0: [HH][ROWP16] halves, then the tap's [64 co][LDP16] slice
    op_t* Bs16 = halo16 + HALO16_H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // Two workgroups share a CU.  They are started together, take the same time and are served round-robin, so they would
    // stay in lockstep: both fill their halo (no MFMAs), both run their taps at half rate, both store (no MFMAs) -- measured:
    // the fill and the epilogue of a tile are not hidden at all (tools/micro/c64_ablate.hip).  A phase offset, once there,
    // persists (each workgroup is replaced when it ends), so the first-round workgroup in the second wave slot of its SIMDs
    // (HW_ID.WAVE_ID odd: tools/micro/hwid_probe.hip) starts about half a tile late.  Placement only changes speed.
    if (p.stagger > 0 && blockIdx.x < 512 && (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1)) {
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    C64_STAMP(0);
#if !(C64_ABL & 32)
    __builtin_amdgcn_s_setprio(3);      // fill / epilogue instructions go ahead of the co-resident workgroup's MFMA stream
#endif
#if C64_ABL & 16
    if (threadIdx.x == 0 && g_c64_trace) g_c64_trace[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    // XCD-aware tile order: workgroups b, b+8, ... share an L2; give each XCD a contiguous run of tiles (neighbouring
    // tiles of an image share halo rows and every tile re-reads the same 147 KB of weights)
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int tx = bid % p.tiles_x;
    const int ty = (bid / p.tiles_x) % p.tiles_y;
    const int n = bid / (p.tiles_x * p.tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const int64_t in_ps = EVAL ? p.in_ps : (int64_t)C;
    const TI* img = (const TI*)p.in + (EVAL ? (int64_t)n * p.in_ss : (int64_t)n * p.H * p.W * C);

    // ---- halo fill: thread -> channel quad c4, pixels q*16 + (tid >> 4) ----
    const int c4 = tid & 15, p0 = tid >> 4;
    constexpr int NPASS = (HH * HW + 15) / 16;     // 12
    f32x4 hv[NPASS];
    bool hin[NPASS];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        const int hp = q * 16 + p0;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        hin[q] = hp < HH * HW && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (hin[q] && !(C64_ABL & 8)) v = ld4(img + ((int64_t)y * p.W + x) * in_ps + c4 * 4);
        hv[q] = v;
    }
    // first weight slice while the halo loads are in flight: thread -> rows (tid >> 4) + 16 i, chunk c4
    f32x4 wv[4];
    const float* wrow = p.wt + (int64_t)(tid >> 4) * 9 * C + c4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[i] = *(const f32x4*)(wrow + (int64_t)i * 16 * 9 * C);
    C64_STAMP(4);                                   // loads issued
    if (!EVAL && p.tr_mean) {
        const f32x4 mu = *(const f32x4*)(p.tr_mean + c4 * 4), is = *(const f32x4*)(p.tr_invstd + c4 * 4);
        const f32x4 ga = *(const f32x4*)(p.tr_gamma + c4 * 4), be = *(const f32x4*)(p.tr_beta + c4 * 4);
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            if (!hin[q]) continue;                 // padding stays exactly zero (it pads the TRANSFORMED activation)
#pragma unroll
            for (int k = 0; k < 4; ++k) hv[q][k] = fmaxf((hv[q][k] - mu[k]) * is[k] * ga[k] + be[k], 0.f);   // bn_apply_fwd's expression
            if (p.emit) {
                const int hp = q * 16 + p0;
                const int hy = hp / HW, hx = hp - hy * HW;
                if (hy >= 1 && hy <= TH && hx >= 1 && hx <= TW)
                    st4((TI*)p.emit + ((int64_t)n * p.H * p.W + (int64_t)(y0 - 1 + hy) * p.W + (x0 - 1 + hx)) * C + c4 * 4, hv[q]);
            }
        }
    }
    auto cvt4 = [](const f32x4& v) { op4 o = {(op_t)v[0], (op_t)v[1], (op_t)v[2], (op_t)v[3]}; return o; };
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        const int hp = q * 16 + p0;
        if (hp < HH * HW) {
            if (OP) *(op4*)(halo16 + (hp / HW) * ROWP16 + (hp % HW) * LDP16 + c4 * 4) = cvt4(hv[q]);
            else *(f32x4*)(halo + (hp / HW) * ROWP + (hp % HW) * LDP + c4 * 4) = hv[q];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (OP) *(op4*)(Bs16 + ((tid >> 4) + 16 * i) * LDP16 + c4 * 4) = cvt4(wv[i]);
        else *(f32x4*)(Bs + ((tid >> 4) + 16 * i) * LDP + c4 * 4) = wv[i];
    }
    C64_STAMP(5);                                   // loads landed, LDS writes done (the stamp waits for them)
    __syncthreads();
    C64_STAMP(1);
#if !(C64_ABL & 32)
    __builtin_amdgcn_s_setprio(0);
#endif

    // ---- main loop: 9 taps x 64 channels; wave w owns tile rows 2w, 2w+1 (32 pixels) x 64 output channels ----
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const float* Ab = halo + (2 * wave + (r >> 4)) * ROWP + (r & 15) * LDP + h * 4;      // tap (0,0) of this lane's pixel
    const float* Bb = Bs + r * LDP + h * 4;

#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - 3 * ky;
        if (t < 8 && !(C64_ABL & 1)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[i] = *(const f32x4*)(wrow + (int64_t)i * 16 * 9 * C + (t + 1) * C);
        }
        if constexpr (OP != 0) {
            // 4 chunks of 16 channels x 2 output-channel halves: lane (r, h) feeds pixel r / filter row r, channels 16 kk + 8 h .. + 7
            const op_t* At16 = halo16 + (2 * wave + (r >> 4) + ky) * ROWP16 + ((r & 15) + kx) * LDP16 + h * 8;
            const op_t* Bb16 = Bs16 + r * LDP16 + h * 8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const op8 a16 = *(const op8*)(At16 + kk * 16);
                const op8 b0 = *(const op8*)(Bb16 + kk * 16), b1 = *(const op8*)(Bb16 + 32 * LDP16 + kk * 16);
                acc[0] = C64Op<OP>::mfma(a16, b0, acc[0]);
                acc[1] = C64Op<OP>::mfma(a16, b1, acc[1]);
            }
            __syncthreads();                         // every wave is done with this tap's weights
            if (t < 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(op4*)(Bs16 + ((tid >> 4) + 16 * i) * LDP16 + c4 * 4) = cvt4(wv[i]);
                __syncthreads();
            }
            continue;
        }
        const float* At = Ab + ky * ROWP + kx * LDP;
        f32x4 a[2], b[2][2];
        a[0] = *(const f32x4*)(At);
        b[0][0] = *(const f32x4*)(Bb);
        b[0][1] = *(const f32x4*)(Bb + 32 * LDP);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int cu = kk & 1, nx = cu ^ 1;
            if (kk + 1 < 8 && !(C64_ABL & 2)) {
                a[nx] = *(const f32x4*)(At + (kk + 1) * 8);
                b[nx][0] = *(const f32x4*)(Bb + (kk + 1) * 8);
                b[nx][1] = *(const f32x4*)(Bb + 32 * LDP + (kk + 1) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);       // keep the fragment reads one chunk ahead of the MFMAs
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = mfma32(a[(C64_ABL & 2) ? 0 : cu][e], b[(C64_ABL & 2) ? 0 : cu][0][e], acc[0]);
                acc[1] = mfma32(a[(C64_ABL & 2) ? 0 : cu][e], b[(C64_ABL & 2) ? 0 : cu][1][e], acc[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (C64_ABL & 1) continue;
        __syncthreads();                             // every wave is done with this tap's weights
        if (t < 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *(f32x4*)(Bs + ((tid >> 4) + 16 * i) * LDP + c4 * 4) = wv[i];
            __syncthreads();
        }
    }

    C64_STAMP(2);
#if !(C64_ABL & 32)
    __builtin_amdgcn_s_setprio(3);
#endif
    // ---- epilogue: statistics straight from the accumulators (pixel = register, channel = lane); the tile goes through
    // LDS (the halo is dead after the last barrier) so that every thread stores -- and reads the residual as -- 16-byte
    // pieces of contiguous NHWC rows, all residual loads in flight at once ----
    // reg e of lane (r, h): pixel row index m = (e & 3) + 8 (e >> 2) + 4 h of the wave's 32, channel j*32 + r
    if (C64_ABL & 4) {
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[0][e] + acc[1][e];
        if (sum == 123.456f) p.out[0] = sum;
        return;
    }
    double s0[2] = {0.0, 0.0}, s1[2] = {0.0, 0.0};
    // ---- direct epilogue (tiles that lie wholly inside the image): accumulator registers straight to HBM.  Register e of lane
    // (r, h) is pixel m = (e & 3) + 8 (e >> 2) + 4 h of the wave's two tile rows, channel j * 32 + r: one wave store writes 2 pixels
    // x 128 contiguous bytes.  No LDS transpose and no barrier -- the instructions outside the tap loop are what the co-resident
    // workgroup's matrix stream cannot hide (profiles/r03_igemm_phases.md).  Training instantiation only: 5.37 -> 5.17 ms per step
    // over its 8 launches; the inference form (16 x 16 maps of 15 979+ patches per launch) lost 2 % with 4-byte stores and keeps
    // the LDS epilogue's 16-byte ones. ----
    if (C64_DIRECT_EPI && !EVAL && !HIO && y0 + TH <= p.H && x0 + TW <= p.W) {
        const int64_t ops_ = EVAL ? p.out_ps : (int64_t)C, rps_ = EVAL ? p.res_ps : (int64_t)C;
        const int64_t obase = (EVAL ? (int64_t)n * p.out_ss : (int64_t)n * p.H * p.W * C) + ((int64_t)(y0 + 2 * wave) * p.W + x0 + 4 * h) * ops_ + r;
        const int64_t rbase = (EVAL ? (int64_t)n * p.res_ss : (int64_t)n * p.H * p.W * C) + ((int64_t)(y0 + 2 * wave) * p.W + x0 + 4 * h) * rps_ + r;
        float scl[2] = {1.f, 1.f}, sft[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (EVAL && p.scale) scl[j] = p.scale[j * 32 + r];
            if (EVAL && p.shift) sft[j] = p.shift[j * 32 + r];
        }
        float rs[16][2];
        if (p.residual) {                           // every residual load in flight before the first use
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pm = (e & 3) + 8 * (e >> 2);                                  // pixel of the lane half h = 0
                const int64_t po = (int64_t)(pm >> 4) * p.W + (pm & 15);
#pragma unroll
                for (int j = 0; j < 2; ++j) rs[e][j] = p.residual[rbase + po * rps_ + j * 32];
            }
            if (!EVAL && p.res_mask) {
                unsigned mk[16][2];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int pm = (e & 3) + 8 * (e >> 2);
                    const int64_t po = (int64_t)(pm >> 4) * p.W + (pm & 15);
#pragma unroll
                    for (int j = 0; j < 2; ++j) mk[e][j] = p.res_mask[(rbase + po * rps_ + j * 32) >> 2];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e)
#pragma unroll
                    for (int j = 0; j < 2; ++j) rs[e][j] = (mk[e][j] >> (r & 3)) & 1u ? rs[e][j] : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) rs[e][0] = rs[e][1] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int pm = (e & 3) + 8 * (e >> 2);
            const int64_t po = (int64_t)(pm >> 4) * p.W + (pm & 15);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float a = acc[j][e];
                if (!EVAL && p.stats) { s0[j] += (double)a; s1[j] += (double)a * (double)a; }
                float v;
                if (EVAL) {                          // the expression of the LDS epilogue / conv_igemm
                    const float x = a * scl[j] + sft[j] + rs[e][j];
                    v = p.relu ? fmaxf(x, 0.f) : x;
                } else {
                    v = a + rs[e][j];
                }
                p.out[obase + po * ops_ + j * 32] = v;
            }
        }
    } else {
    float* Ct = halo;                               // [128 pixels][LDP]
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
        const int y = y0 + 2 * wave + (m >> 4), x = x0 + (m & 15);
        const bool ok = y < p.H && x < p.W;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float v = (p.stats && !p.residual) ? stored<TI>(acc[j][e]) : acc[j][e];      // statistics of what is stored
            if (!EVAL && p.stats && ok) { s0[j] += (double)v; s1[j] += (double)v * (double)v; }
            Ct[(wave * 32 + m) * LDP + j * 32 + r] = v;
        }
    }
    __syncthreads();
    {
        // thread -> channel quad c4, tile pixels (tid >> 4) + 16 q
        int64_t o[8];
        f32x4 rs[8];
        f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
        if (EVAL && p.scale) sc4 = *(const f32x4*)(p.scale + c4 * 4);
        if (EVAL && p.shift) sh4 = *(const f32x4*)(p.shift + c4 * 4);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int tp = (tid >> 4) + 16 * q;
            const int y = y0 + (tp >> 4), x = x0 + (tp & 15);
            const bool ok = y < p.H && x < p.W;
            if (EVAL) o[q] = ok ? (int64_t)n * p.out_ss + ((int64_t)y * p.W + x) * p.out_ps + c4 * 4 : -1;
            else o[q] = ok ? (((int64_t)n * p.H + y) * p.W + x) * C + c4 * 4 : -1;
            f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            if (EVAL) rs[q] = (ok && p.residual) ? *(const f32x4*)(p.residual + (int64_t)n * p.res_ss + ((int64_t)y * p.W + x) * p.res_ps + c4 * 4) : z4;
            else rs[q] = (ok && p.residual) ? ld4((const TI*)p.residual + o[q]) : z4;
            if (!EVAL && ok && p.res_mask) {
                const unsigned mk = p.res_mask[o[q] >> 2];
#pragma unroll
                for (int k = 0; k < 4; ++k) rs[q][k] = (mk >> k) & 1u ? rs[q][k] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int tp = (tid >> 4) + 16 * q;
            f32x4 v = *(const f32x4*)(Ct + tp * LDP + c4 * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (EVAL) {                         // the expression of conv_igemm's epilogue
                    const float x = v[k] * sc4[k] + sh4[k] + rs[q][k];
                    v[k] = p.relu ? fmaxf(x, 0.f) : x;
                } else {
                    v[k] += rs[q][k];
                }
            }
            if (o[q] >= 0) st4((TI*)p.out + o[q], v);
        }
    }
    }
    C64_STAMP(3);
    if (!EVAL && p.stats) {
        // lane halves -> one value per channel per wave, then the four waves in a fixed order through LDS
        double* S = OP ? (double*)Bs16 : (double*)Bs;      // [4 waves][2][64]; the weight buffer is free (barrier above)
        if (OP) __syncthreads();                     // ... but in the 16-bit layout it lies inside the LDS epilogue's tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s0[j] += __shfl_xor(s0[j], 32);
            s1[j] += __shfl_xor(s1[j], 32);
            if (h == 0) {
                S[(wave * 2 + 0) * C + j * 32 + r] = s0[j];
                S[(wave * 2 + 1) * C + j * 32 + r] = s1[j];
            }
        }
        __syncthreads();
        if (tid < 2 * C) {
            const int which = tid / C, cc = tid % C;
            const double t = ((S[(0 * 2 + which) * C + cc] + S[(1 * 2 + which) * C + cc]) + S[(2 * 2 + which) * C + cc]) +
                             S[(3 * 2 + which) * C + cc];
            p.stats[((int64_t)blockIdx.x * 2 + which) * C + cc] = t;
        }
    }
}

// start offset between the two workgroups of a CU (see the kernel), only when the launch has several rounds of them
static int c64_stagger(int64_t nwg) {
    static const int st = getenv("SSAD_C64_STAGGER") ? atoi(getenv("SSAD_C64_STAGGER")) : 0;
    return nwg >= 4 * 512 ? st : 0;
}

}  // namespace

extern "C" int64_t ssad_conv3x3_c64_stats_rows(int64_t N, int H, int W) {
    return N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
}

// out = conv3x3(pad 1, stride 1)(T(in)) (+ residual [* res_mask]), 64 -> 64 channels, NHWC fp32, OHWI weights.
// res_mask (optional): nibble mask of ssad_bn_apply_fwd_mask -- the residual is the gradient of a ReLU output and only its
// active elements flow into the identity branch (the masked copy is never materialised).
// T = identity, or relu((x - tr_mean) * tr_invstd * tr_gamma + tr_beta) per input channel when tr_mean != NULL (then
// `emit`, if given, receives T(in): the activation the weight-gradient kernel of this layer needs).
// stats_ws != NULL: train-mode BatchNorm statistics of the output (ssad_conv3x3_c64_stats_rows(N, H, W) * 2 * 64 doubles
// of workspace), finalised exactly as ssad_conv_igemm_fwd_stats does.
// op: 0 exact fp32, 1 bf16 operands, 2 fp16 operands (fp32 accumulation; csrc comment at the kernel)
static int conv3x3_c64_impl(const float* in, const float* w_ohwi, float* out, const float* residual, const uint8_t* res_mask,
                            const float* tr_mean, const float* tr_invstd, const float* tr_gamma, const float* tr_beta, float* emit,
                            int64_t N, int H, int W, double* stats_ws, float eps, float momentum, float* mean, float* invstd,
                            float* running_mean, float* running_var, int op, int half_io, void* stream) {
    SSAD_CHECK_ARG(in && w_ohwi && out && N > 0 && H > 0 && W > 0, "bad argument");
    SSAD_CHECK_ARG(op >= 0 && op <= 2, "op: 0 fp32, 1 bf16, 2 fp16");
    SSAD_CHECK_ARG(!half_io || (op == 2 && !res_mask), "half tensors: fp16 operands, no residual mask");
    SSAD_CHECK_ARG(!tr_mean || (tr_invstd && tr_gamma && tr_beta), "input transform needs mean, invstd, gamma, beta");
    SSAD_CHECK_ARG(!emit || tr_mean, "emit without an input transform");
    SSAD_CHECK_ARG(!stats_ws || (mean && invstd), "statistics need mean / invstd outputs");
    C64Params p;
    SSAD_CHECK_ARG(!res_mask || residual, "residual mask without a residual");
    p.in = in; p.wt = w_ohwi; p.out = out; p.residual = residual; p.res_mask = res_mask;
    p.tr_mean = tr_mean; p.tr_invstd = tr_invstd; p.tr_gamma = tr_gamma; p.tr_beta = tr_beta; p.emit = emit;
    p.stats = stats_ws;
    p.scale = p.shift = nullptr; p.relu = 0; p.in_ps = p.out_ps = p.res_ps = C; p.in_ss = p.out_ss = p.res_ss = (int64_t)H * W * C;
    p.N = (int)N; p.H = H; p.W = W;
    p.tiles_y = (H + TH - 1) / TH; p.tiles_x = (W + TW - 1) / TW;
    const int64_t nwg = N * p.tiles_y * p.tiles_x;
    SSAD_CHECK_ARG(nwg < (int64_t)2147483647, "too many tiles for one launch");
    p.stagger = c64_stagger(nwg);
    static const int lds_bytes = LDS_BYTES + (getenv("SSAD_C64_LDS_PAD") ? atoi(getenv("SSAD_C64_LDS_PAD")) : 0);   // residency experiments
    static bool attr_set = false;
    if (!attr_set) {
        SSAD_SET_DYN_LDS(conv3x3_c64_kernel<false>, lds_bytes);
        attr_set = true;
    }
    if (half_io) hipLaunchKernelGGL((conv3x3_c64_kernel<false, 2, hf>), dim3((unsigned)nwg), dim3(256), LDS_BYTES16, (hipStream_t)stream, p);
    else if (op == 2) hipLaunchKernelGGL((conv3x3_c64_kernel<false, 2>), dim3((unsigned)nwg), dim3(256), LDS_BYTES16, (hipStream_t)stream, p);
    else if (op == 1) hipLaunchKernelGGL((conv3x3_c64_kernel<false, 1>), dim3((unsigned)nwg), dim3(256), LDS_BYTES16, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv3x3_c64_kernel<false>, dim3((unsigned)nwg), dim3(256), lds_bytes, (hipStream_t)stream, p);
    SSAD_CHECK_LAUNCH();
    if (stats_ws)
        return ssad_bn_finalize_partials(stats_ws, (int)nwg, N * H * W, C, eps, momentum, mean, invstd, running_mean,
                                         running_var, stream);
    return 0;
}

extern "C" int ssad_conv3x3_c64_op(const float* in, const float* w_ohwi, float* out, const float* residual, const uint8_t* res_mask,
                                   const float* tr_mean, const float* tr_invstd, const float* tr_gamma, const float* tr_beta, float* emit,
                                   int64_t N, int H, int W, double* stats_ws, float eps, float momentum, float* mean, float* invstd,
                                   float* running_mean, float* running_var, int op, void* stream) {
    return conv3x3_c64_impl(in, w_ohwi, out, residual, res_mask, tr_mean, tr_invstd, tr_gamma, tr_beta, emit, N, H, W, stats_ws, eps,
                            momentum, mean, invstd, running_mean, running_var, op, 0, stream);
}

// in / out / residual / emit stored as halves, fp16 operands (the weights stay the fp32 OHWI master copy, rounded while staged):
// layer1 of the precision-16 step with half tensors
extern "C" int ssad_conv3x3_c64_h(const void* in, const float* w_ohwi, void* out, const void* residual,
                                  const float* tr_mean, const float* tr_invstd, const float* tr_gamma, const float* tr_beta, void* emit,
                                  int64_t N, int H, int W, double* stats_ws, float eps, float momentum, float* mean, float* invstd,
                                  float* running_mean, float* running_var, void* stream) {
    return conv3x3_c64_impl((const float*)in, w_ohwi, (float*)out, (const float*)residual, nullptr, tr_mean, tr_invstd, tr_gamma, tr_beta,
                            (float*)emit, N, H, W, stats_ws, eps, momentum, mean, invstd, running_mean, running_var, 2, 1, stream);
}

extern "C" int ssad_conv3x3_c64(const float* in, const float* w_ohwi, float* out, const float* residual, const uint8_t* res_mask, const float* tr_mean,
                                const float* tr_invstd, const float* tr_gamma, const float* tr_beta, float* emit, int64_t N,
                                int H, int W, double* stats_ws, float eps, float momentum, float* mean, float* invstd,
                                float* running_mean, float* running_var, void* stream) {
    return ssad_conv3x3_c64_op(in, w_ohwi, out, residual, res_mask, tr_mean, tr_invstd, tr_gamma, tr_beta, emit, N, H, W, stats_ws, eps,
                               momentum, mean, invstd, running_mean, running_var, 0, stream);
}

// Inference form (frozen BatchNorm folded into scale / shift): out = act(conv3x3(in) * scale[co] + shift[co] + residual),
// 64 -> 64 channels, stride 1, pad 1.  in_hwnc / out_hwnc / res_hwnc select the position-major [H][W][N][64] layout of the
// patch-scoring trunk for the input / the output / the residual (0 = NHWC).  Replaces ssad_conv_igemm_fwd(_hwnc) for
// the four layer1 convolutions of a scoring pass (same call sites: torchvision BasicBlock under PeraNet.forward,
// models.py:224 of the reference): one halo load per tile instead of one gather per filter tap.
extern "C" int ssad_conv3x3_c64_eval(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                                     const float* residual, int relu, int64_t N, int H, int W, int in_hwnc, int out_hwnc,
                                     int res_hwnc, void* stream) {
    SSAD_CHECK_ARG(in && w_ohwi && out && N > 0 && H > 0 && W > 0, "bad argument");
    C64Params p;
    p.in = in; p.wt = w_ohwi; p.out = out; p.residual = residual; p.res_mask = nullptr;
    p.tr_mean = p.tr_invstd = p.tr_gamma = p.tr_beta = nullptr; p.emit = nullptr; p.stats = nullptr;
    p.scale = scale; p.shift = shift; p.relu = relu;
    p.in_ps = in_hwnc ? N * C : C;   p.in_ss = in_hwnc ? C : (int64_t)H * W * C;
    p.out_ps = out_hwnc ? N * C : C; p.out_ss = out_hwnc ? C : (int64_t)H * W * C;
    p.res_ps = res_hwnc ? N * C : C; p.res_ss = res_hwnc ? C : (int64_t)H * W * C;
    p.N = (int)N; p.H = H; p.W = W;
    p.tiles_y = (H + TH - 1) / TH; p.tiles_x = (W + TW - 1) / TW;
    const int64_t nwg = N * p.tiles_y * p.tiles_x;
    SSAD_CHECK_ARG(nwg < (int64_t)2147483647, "too many tiles for one launch");
    p.stagger = c64_stagger(nwg);
    static bool attr_set = false;
    if (!attr_set) {
        SSAD_SET_DYN_LDS(conv3x3_c64_kernel<true>, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3x3_c64_kernel<true>, dim3((unsigned)nwg), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    SSAD_CHECK_LAUNCH();
    return 0;
}
