// HBM-bound helpers: weight repacking, global average pool, cosine 3-NN reduction, blur+ReLU+bilinear.
#include "common.h"
#include <stdarg.h>

// ---------------------------------------------------------------------------------------------
// error reporting
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void ssad_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ssad_last_error(void) { return g_err; }
extern "C" int ssad_version(void) { return 100; }

namespace {

// OIHW <-> OHWI (one thread per element; weights are tiny next to activations)
__global__ void repack_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int I, int KH, int KW,
                              int to_ohwi) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)O * I * KH * KW;
    if (idx >= total) return;
    // idx enumerates OHWI
    int ci = (int)(idx % I);
    int64_t t = idx / I;
    int kx = (int)(t % KW); t /= KW;
    int ky = (int)(t % KH);
    int o = (int)(t / KH);
    int64_t oihw = (((int64_t)o * I + ci) * KH + ky) * KW + kx;
    if (to_ohwi) dst[idx] = src[oihw];
    else dst[oihw] = src[idx];
}

// in [N][HW][C] -> out[n*stride + off + c] = sum/HW.  One thread per (n, c); lanes run over c (coalesced).
__global__ void gap_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t N, int HW, int C,
                           int out_stride, int out_offset, int hwnc) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    int c = (int)(i % C);
    int64_t n = i / C;
    const float* p = hwnc ? in + n * C + c : in + n * HW * C + c;
    const int64_t step = hwnc ? N * C : (int64_t)C;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= HW; k += 8) {          // eight loads in flight, added in position order (the sum is unchanged)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(k + j) * step];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; k < HW; ++k) s += p[k * step];
    out[n * out_stride + out_offset + c] = s / (float)HW;
}

// Few samples, large maps (training batches): one workgroup per (sample, 64-channel slab); 16 pixel lanes x 16 float4
// channel lanes, lane sums added in a fixed order through LDS.
template <typename T>
__global__ __launch_bounds__(256) void gap_wide_kernel(const T* __restrict__ in, float* __restrict__ out, int HW, int C,
                                                       int out_stride, int out_offset) {
    __shared__ f32x4 part[16][16];
    const int cq = threadIdx.x & 15, pg = threadIdx.x >> 4;
    const int64_t n = blockIdx.x;
    const int c0 = blockIdx.y * 64 + cq * 4;
    const T* p = in + n * HW * C + c0;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = pg; k < HW; k += 16) {
        const f32x4 v = ld4(p + (int64_t)k * C);
        s += v;
    }
    part[pg][cq] = s;
    __syncthreads();
    if (pg == 0) {
        f32x4 t = part[0][cq];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += part[q][cq];
        float* o = out + n * out_stride + out_offset + c0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = t[k] / (float)HW;
    }
}

// one wave per row
__global__ void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t N, int D) {
    int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= N) return;
    int lane = threadIdx.x & 63;
    const float* p = x + row * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s += p[k] * p[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    float nrm = sqrtf(s);
    for (int k = lane; k < D; k += 64) out[row * D + k] = p[k] / nrm;
}

__device__ __forceinline__ void insert3(float v, float& a, float& b, float& c) {
    // keeps a <= b <= c as the three smallest
    if (v < c) {
        if (v < b) {
            c = b;
            if (v < a) { b = a; a = v; } else b = v;
        } else c = v;
    }
}

// sim [Nq][Nb] -> mean of the k (<=3) smallest clip(1-sim,0,2).  One wave per row.
__global__ void knn_mean_kernel(const float* __restrict__ sim, float* __restrict__ out, int64_t Nq, int Nb, int k) {
    int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= Nq) return;
    int lane = threadIdx.x & 63;
    const float* p = sim + row * Nb;
    float a = INFINITY, b = INFINITY, c = INFINITY;
    for (int j = lane; j < Nb; j += 64) {
        float d = 1.f - p[j];
        d = fminf(fmaxf(d, 0.f), 2.f);
        insert3(d, a, b, c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float oa = __shfl_xor(a, o), ob = __shfl_xor(b, o), oc = __shfl_xor(c, o);
        insert3(oa, a, b, c);
        insert3(ob, a, b, c);
        insert3(oc, a, b, c);
    }
    if (lane == 0) {
        float s = a;
        if (k > 1) s += b;
        if (k > 2) s += c;
        out[row] = s / (float)k;
    }
}

// Grad-CAM channel contraction: out[b][pos] = sum_k alpha[b][k] * act[b][pos][k]   (act NHWC).  One wave per position;
// lane l adds channels l, l+64, ... in order, fixed butterfly across lanes.
__global__ void gradcam_map_kernel(const float* __restrict__ act, const float* __restrict__ alpha, float* __restrict__ out,
                                   int64_t rows, int HW, int C, int alpha_stride) {
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* a = act + row * C;
    const float* w = alpha + (row / HW) * alpha_stride;
    float s = 0.f;
    for (int k = lane; k < C; k += 64) s += a[k] * w[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = s;
}

// relu(gaussian_blur(reflect pad)) then bilinear(align_corners=False).  One workgroup per map.
__global__ void blur_relu_bilinear_kernel(const float* __restrict__ maps, float* __restrict__ out, int h, int w, int ks,
                                          int target) {
    extern __shared__ float sm[];
    float* src = sm;                 // h*w
    float* blr = sm + h * w;         // h*w
    float* k1 = blr + h * w;         // ks
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* m = maps + (int64_t)n * h * w;
    for (int i = tid; i < h * w; i += blockDim.x) src[i] = m[i];
    if (tid == 0) {
        // torchvision: sigma = 0.15*k + 0.35; x = linspace(-(k-1)/2, (k-1)/2, k); pdf = exp(-0.5 (x/sigma)^2); k1 = pdf/sum
        float sigma = 0.15f * (float)ks + 0.35f;
        float half = (float)(ks - 1) * 0.5f, sum = 0.f;
        for (int i = 0; i < ks; ++i) {
            float x = -half + (float)i;
            float v = expf(-0.5f * (x / sigma) * (x / sigma));
            k1[i] = v;
            sum += v;
        }
        for (int i = 0; i < ks; ++i) k1[i] /= sum;
    }
    __syncthreads();
    const int pad = ks / 2;
    for (int i = tid; i < h * w; i += blockDim.x) {
        int y = i / w, x = i - y * w;
        float acc = 0.f;
        for (int dy = 0; dy < ks; ++dy) {
            int yy = y + dy - pad;
            yy = yy < 0 ? -yy : (yy >= h ? 2 * h - 2 - yy : yy);
            for (int dx = 0; dx < ks; ++dx) {
                int xx = x + dx - pad;
                xx = xx < 0 ? -xx : (xx >= w ? 2 * w - 2 - xx : xx);
                acc += src[yy * w + xx] * (k1[dy] * k1[dx]);
            }
        }
        blr[i] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    const float shy = (float)h / (float)target, swx = (float)w / (float)target;
    float* o = out + (int64_t)n * target * target;
    for (int i = tid; i < target * target; i += blockDim.x) {
        int y = i / target, x = i - y * target;
        float sy = fmaxf(shy * ((float)y + 0.5f) - 0.5f, 0.f);
        float sx = fmaxf(swx * ((float)x + 0.5f) - 0.5f, 0.f);
        int y0 = (int)sy, x0 = (int)sx;
        int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        float ly = sy - (float)y0, lx = sx - (float)x0;
        float hy = 1.f - ly, hx = 1.f - lx;
        o[i] = hy * (hx * blr[y0 * w + x0] + lx * blr[y0 * w + x1]) + ly * (hx * blr[y1 * w + x0] + lx * blr[y1 * w + x1]);
    }
}

// The same map -> map function for maps that do not fit one workgroup's LDS (WideResNet-50 scales: 128 x 128 -> 512 x 512): one
// workgroup per TT x TT tile of the OUTPUT; it stages the source rows / columns its bilinear taps touch plus the blur halo (reflect
// indices resolved while loading), blurs them in LDS and interpolates.  Same expressions, same summation order per pixel as the
// single-workgroup kernel: identical results.
constexpr int BLUR_TT = 64;
__global__ void blur_relu_bilinear_tiled_kernel(const float* __restrict__ maps, float* __restrict__ out, int h, int w, int ks,
                                                int target, int tiles, int rmax) {
    extern __shared__ float sm[];
    const int pad = ks / 2, sw = rmax + 2 * pad;
    float* src = sm;                      // [rmax + 2 pad][rmax + 2 pad]
    float* blr = sm + sw * sw;            // [rmax][rmax]
    float* k1 = blr + rmax * rmax;        // ks
    const int n = blockIdx.y, tid = threadIdx.x;
    const int ty0 = (blockIdx.x / tiles) * BLUR_TT, tx0 = (blockIdx.x % tiles) * BLUR_TT;
    const int ty1 = min(ty0 + BLUR_TT, target) - 1, tx1 = min(tx0 + BLUR_TT, target) - 1;
    const float shy = (float)h / (float)target, swx = (float)w / (float)target;
    const int ya = (int)fmaxf(shy * ((float)ty0 + 0.5f) - 0.5f, 0.f), xa = (int)fmaxf(swx * ((float)tx0 + 0.5f) - 0.5f, 0.f);
    const int yb = min((int)fmaxf(shy * ((float)ty1 + 0.5f) - 0.5f, 0.f) + 1, h - 1);
    const int xb = min((int)fmaxf(swx * ((float)tx1 + 0.5f) - 0.5f, 0.f) + 1, w - 1);
    const int nr = yb - ya + 1, nc = xb - xa + 1;               // <= rmax by construction (host)
    const float* m = maps + (int64_t)n * h * w;
    for (int i = tid; i < (nr + 2 * pad) * (nc + 2 * pad); i += blockDim.x) {
        const int ly = i / (nc + 2 * pad), lx = i - ly * (nc + 2 * pad);
        int yy = ya - pad + ly, xx = xa - pad + lx;
        yy = yy < 0 ? -yy : (yy >= h ? 2 * h - 2 - yy : yy);
        xx = xx < 0 ? -xx : (xx >= w ? 2 * w - 2 - xx : xx);
        src[ly * sw + lx] = m[yy * w + xx];
    }
    if (tid == 0) {
        float sigma = 0.15f * (float)ks + 0.35f;
        float half = (float)(ks - 1) * 0.5f, sum = 0.f;
        for (int i = 0; i < ks; ++i) {
            float x = -half + (float)i;
            float v = expf(-0.5f * (x / sigma) * (x / sigma));
            k1[i] = v;
            sum += v;
        }
        for (int i = 0; i < ks; ++i) k1[i] /= sum;
    }
    __syncthreads();
    for (int i = tid; i < nr * nc; i += blockDim.x) {
        const int ly = i / nc, lx = i - ly * nc;
        float acc = 0.f;
        for (int dy = 0; dy < ks; ++dy)
            for (int dx = 0; dx < ks; ++dx) acc += src[(ly + dy) * sw + lx + dx] * (k1[dy] * k1[dx]);
        blr[ly * rmax + lx] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    float* o = out + (int64_t)n * target * target;
    const int th = ty1 - ty0 + 1, tw = tx1 - tx0 + 1;
    for (int i = tid; i < th * tw; i += blockDim.x) {
        const int y = ty0 + i / tw, x = tx0 + i % tw;
        float sy = fmaxf(shy * ((float)y + 0.5f) - 0.5f, 0.f);
        float sx = fmaxf(swx * ((float)x + 0.5f) - 0.5f, 0.f);
        int y0 = (int)sy, x0 = (int)sx;
        int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        float ly = sy - (float)y0, lx = sx - (float)x0;
        float hy = 1.f - ly, hx = 1.f - lx;
        y0 -= ya; y1 -= ya; x0 -= xa; x1 -= xa;
        o[(int64_t)y * target + x] = hy * (hx * blr[y0 * rmax + x0] + lx * blr[y0 * rmax + x1]) +
                                     ly * (hx * blr[y1 * rmax + x0] + lx * blr[y1 * rmax + x1]);
    }
}

}  // namespace

static int repack(const float* src, float* dst, int O, int I, int KH, int KW, int to_ohwi, void* stream) {
    int64_t total = (int64_t)O * I * KH * KW;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, O, I,
                       KH, KW, to_ohwi);
    return 0;
}

extern "C" int ssad_repack_oihw_to_ohwi(const float* w_oihw, float* w_ohwi, int O, int I, int KH, int KW, void* stream) {
    SSAD_CHECK_ARG(w_oihw && w_ohwi && O > 0 && I > 0 && KH > 0 && KW > 0, "bad argument");
    repack(w_oihw, w_ohwi, O, I, KH, KW, 1, stream);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_repack_ohwi_to_oihw(const float* w_ohwi, float* w_oihw, int O, int I, int KH, int KW, void* stream) {
    SSAD_CHECK_ARG(w_oihw && w_ohwi && O > 0 && I > 0 && KH > 0 && KW > 0, "bad argument");
    repack(w_ohwi, w_oihw, O, I, KH, KW, 0, stream);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_gap_fwd(const float* in, float* out, int64_t N, int HW, int C, int out_stride, int out_offset,
                            int hwnc, void* stream) {
    SSAD_CHECK_ARG(in && out && N > 0 && HW > 0 && C > 0, "bad argument");
    SSAD_CHECK_ARG(out_offset >= 0 && out_offset + C <= out_stride, "slice does not fit the output row");
    if (!hwnc && C % 64 == 0 && HW >= 64 && N * C < 256 * 1024)
        hipLaunchKernelGGL(gap_wide_kernel<float>, dim3((unsigned)N, C / 64), dim3(256), 0, (hipStream_t)stream, in, out, HW, C,
                           out_stride, out_offset);
    else
        hipLaunchKernelGGL(gap_kernel, dim3((unsigned)cdiv64(N * C, 256)), dim3(256), 0, (hipStream_t)stream, in, out, N, HW, C,
                           out_stride, out_offset, hwnc);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// in stored as halves [N][HW][C] (C % 64 == 0: the precision-16 training trunk), out fp32 as above
extern "C" int ssad_gap_fwd_h(const void* in, float* out, int64_t N, int HW, int C, int out_stride, int out_offset, void* stream) {
    SSAD_CHECK_ARG(in && out && N > 0 && HW > 0 && C > 0 && C % 64 == 0, "bad argument (C % 64)");
    SSAD_CHECK_ARG(out_offset >= 0 && out_offset + C <= out_stride, "slice does not fit the output row");
    hipLaunchKernelGGL(gap_wide_kernel<hf>, dim3((unsigned)N, C / 64), dim3(256), 0, (hipStream_t)stream, (const hf*)in, out, HW, C,
                       out_stride, out_offset);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// ---- interior positions of a patch's map, copied from the per-image dense map (patch scoring, layer1) ----
// out is position-major [H][W][N][C], N = B * prow * pcol patches, patch n = (b, pr, pc) in the order of extract_patches
// (src/self_supervised/functional.py:77-82: p = pr * pcol + pc); dense is NHWC [B][Hd][Wd][C].  For every position lo <= u, v <= hi:
// out[u][v][n][:] = dense[b][shift * pr + u][shift * pc + v][:] -- the value the patch-wise conv would compute there, because no
// zero-padded border of the patch is within reach of that position (ssad_conv_igemm_fwd_hwnc_ring computes the others).
namespace {
__global__ __launch_bounds__(256) void patch_gather_hwnc_kernel(const f32x4* __restrict__ dense, f32x4* __restrict__ out, int64_t N, int prow,
                                                               int pcol, int shift, int Hd, int Wd, int C4, int W, int lo, int side,
                                                               int ilo, int iside) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C4) return;
    const int c4 = (int)(i % C4);
    const int64_t n = i / C4;
    const int pp = prow * pcol;
    const int64_t b = n / pp;
    const int rem = (int)(n - b * pp);
    const int pr = rem / pcol, pc = rem - pr * pcol;
    // blockIdx.y counts the positions of the square [lo, lo + side) that lie OUTSIDE the inner square [ilo, ilo + iside) (iside = 0:
    // none), row by row: full rows above, two flanks per row beside, full rows below
    int u, v;
    {
        const int k = (int)blockIdx.y, top = (ilo - lo) * side, w2 = side - iside, mid = iside * w2;
        if (iside == 0 || k < top) { u = lo + k / side; v = lo + k % side; }
        else if (k < top + mid) {
            const int kk = k - top, j = kk % w2;
            u = ilo + kk / w2;
            v = j < ilo - lo ? lo + j : ilo + iside + (j - (ilo - lo));
        } else { const int kk = k - top - mid; u = ilo + iside + kk / side; v = lo + kk % side; }
    }
    const f32x4 val = dense[((b * Hd + shift * pr + u) * Wd + shift * pc + v) * C4 + c4];
    out[(((int64_t)u * W + v) * N + n) * C4 + c4] = val;
}
}  // namespace

// ... of the square lo <= u, v <= hi only the positions outside the inner square ilo <= u, v <= ihi (ilo > ihi: the whole square): the
// next ring conv reads its input within one position of the outputs it computes, so the deep interior of an intermediate map is never
// read by anyone (round 6: 83 of the 276 positions copied per image pass).
extern "C" int ssad_patch_gather_hwnc_band(const float* dense, float* out, int64_t B, int prow, int pcol, int shift, int Hd, int Wd, int C,
                                           int H, int W, int lo, int hi, int ilo, int ihi, void* stream) {
    SSAD_CHECK_ARG(dense && out && B > 0 && prow > 0 && pcol > 0 && shift > 0 && C > 0 && C % 4 == 0, "bad argument");
    SSAD_CHECK_ARG(lo >= 0 && hi >= lo && hi < H && hi < W, "the copied square must lie inside the map");
    SSAD_CHECK_ARG(ilo > ihi || (ilo > lo && ihi < hi), "the inner square must lie strictly inside the copied one");
    SSAD_CHECK_ARG(shift * (prow - 1) + hi < Hd && shift * (pcol - 1) + hi < Wd, "the dense map does not cover the last patch");
    const int64_t N = B * prow * pcol;
    const int side = hi - lo + 1;
    const int iside = ilo > ihi ? 0 : ihi - ilo + 1;
    SSAD_CHECK_ARG(cdiv64(N * (C / 4), 256) < (int64_t)2147483647 && side * side <= 65535, "too large for one launch");
    hipLaunchKernelGGL(patch_gather_hwnc_kernel, dim3((unsigned)cdiv64(N * (C / 4), 256), side * side - iside * iside), dim3(256), 0,
                       (hipStream_t)stream, (const f32x4*)dense, (f32x4*)out, N, prow, pcol, shift, Hd, Wd, C / 4, W, lo, side, ilo, iside);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_patch_gather_hwnc(const float* dense, float* out, int64_t B, int prow, int pcol, int shift, int Hd, int Wd, int C,
                                      int H, int W, int lo, int hi, void* stream) {
    return ssad_patch_gather_hwnc_band(dense, out, B, prow, pcol, shift, Hd, Wd, C, H, W, lo, hi, 1, 0, stream);
}

extern "C" int ssad_gradcam_map(const float* act, const float* alpha, float* out, int64_t B, int HW, int C, int alpha_stride,
                                void* stream) {
    SSAD_CHECK_ARG(act && alpha && out && B > 0 && HW > 0 && C > 0 && alpha_stride >= C, "bad argument");
    const int64_t rows = B * HW;
    hipLaunchKernelGGL(gradcam_map_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, act, alpha, out,
                       rows, HW, C, alpha_stride);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_l2_normalize_rows(const float* x, float* out, int64_t N, int D, void* stream) {
    SSAD_CHECK_ARG(x && out && N > 0 && D > 0, "bad argument");
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)cdiv64(N, 4)), dim3(256), 0, (hipStream_t)stream, x, out, N, D);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_cosine_knn_mean(const float* sim, float* out, int64_t Nq, int Nb, int k, void* stream) {
    SSAD_CHECK_ARG(sim && out && Nq > 0 && Nb > 0, "bad argument");
    SSAD_CHECK_ARG(k >= 1 && k <= 3 && k <= Nb, "k must be 1..3 and <= bank rows");
    hipLaunchKernelGGL(knn_mean_kernel, dim3((unsigned)cdiv64(Nq, 4)), dim3(256), 0, (hipStream_t)stream, sim, out, Nq, Nb, k);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_blur_relu_bilinear(const float* maps, float* out, int n, int h, int w, int ksize, int target,
                                       void* stream) {
    SSAD_CHECK_ARG(maps && out && n > 0 && h > 0 && w > 0 && target > 0, "bad argument");
    SSAD_CHECK_ARG(ksize >= 1 && (ksize & 1) && ksize / 2 < h && ksize / 2 < w, "kernel must be odd and reflect-pad must fit");
    size_t lds = (size_t)(2 * h * w + ksize) * sizeof(float);
    if (lds <= 48 * 1024) {
        hipLaunchKernelGGL(blur_relu_bilinear_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, maps, out, h, w, ksize, target);
        SSAD_CHECK_LAUNCH();
        return 0;
    }
    // larger maps: one workgroup per 64 x 64 tile of the output
    const int tiles = (target + BLUR_TT - 1) / BLUR_TT;
    const int span = BLUR_TT < target ? BLUR_TT : target;
    const int rmax_y = (int)((double)h / target * span) + 3, rmax_x = (int)((double)w / target * span) + 3;
    const int rmax = rmax_y > rmax_x ? rmax_y : rmax_x;
    const int sw = rmax + 2 * (ksize / 2);
    lds = (size_t)(sw * sw + rmax * rmax + ksize) * sizeof(float);
    SSAD_CHECK_ARG(lds <= 64 * 1024, "down-sampling ratio too large for the tiled kernel");
    hipLaunchKernelGGL(blur_relu_bilinear_tiled_kernel, dim3(tiles * tiles, n), dim3(256), lds, (hipStream_t)stream, maps, out, h, w,
                       ksize, target, tiles, rmax);
    SSAD_CHECK_LAUNCH();
    return 0;
}
