// obj_mask on the device: Canny edges -> dilate -> close -> fill holes -> erode -> largest component.
//
// Replaces dataset_generator.obj_mask of the reference (src/self_supervised/dataset_generator.py:27-39: skimage.feature.canny(sigma 1.5,
// thresholds 5 / 15) + skimage.morphology / scipy.ndimage), which it runs on the HOST for every sample of the non-fixed-object
// categories (datasets.py:226) -- here once per image of a category when the GPU-resident loader is built, for the whole batch of
// uint8 RGB images that already sits in HBM after the device-side resize (csrc/resize.hip).
//
// Bit-exact by construction against the host statement (self_supervised/dataset_generator._canny / obj_mask, itself pinned to
// scikit-image 0.18.3 by tests/golden/skimage.npz):
//   * Pillow's integer luma; the [0, 1] image, Gaussian smoothing, Sobel derivatives, magnitude, non-maximum suppression and the
//     thresholds in fp64 with scipy.ndimage's operation ORDER (correlate1d's symmetric / antisymmetric pair sums, outermost pair
//     first; 'constant' borders for the Gaussian, 'reflect' for Sobel) and no fused multiply-add (this file is compiled with
//     -ffp-contract=off, like augment.hip); the Gaussian weights come from the host (numpy's exp, as scipy computes them);
//   * np.hypot == glibc 2.35's hypot (not correctly rounded in ~0.2 % of inputs): restated operation by operation (hypot64);
//   * hysteresis, hole filling and component labelling are fixed points of monotone propagations (the result does not depend on the
//     order of updates): iterated inside one workgroup per image until nothing changes; component sizes by integer atomics;
//     ties go to the component whose first pixel comes first in raster order, as np.argmax over scipy's labels does.
// HBM-light (a 256 x 256 image is 2 MB of doubles) and latency-bound: a few hundred microseconds per batch, once per category.
#include "common.h"

namespace {

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// ---- element-wise / stencil passes over [B][H][W] doubles: one thread per pixel ----
__global__ void om_gray_kernel(const uint8_t* __restrict__ rgb, double* __restrict__ out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    out[i] = (double)luma(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]) / 255.0;
}

// scipy.ndimage.correlate1d with a SYMMETRIC kernel w[0 .. 2r] along `axis` (0 = rows, 1 = columns), mode 'constant' (cval 0):
// tmp = x[i] * w[r]; for j = -r .. -1: tmp += (x[i + j] + x[i - j]) * w[r + j]      (ni_filters.c, NI_Correlate1D)
// ones != 0: the input is the all-ones image (the `bleed` normaliser of skimage's canny)
__global__ void om_gauss_kernel(const double* __restrict__ in, double* __restrict__ out, int B, int H, int W, int axis,
                                const double* __restrict__ w, int r, int ones) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)B * H * W;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int p = axis ? x : y, n = axis ? W : H;
    const int64_t step = axis ? 1 : W;
    auto at = [&](int q) -> double { return (q < 0 || q >= n) ? 0.0 : (ones ? 1.0 : in[i + (int64_t)(q - p) * step]); };
    double tmp = at(p) * w[r];
    for (int j = -r; j < 0; ++j) tmp += (at(p + j) + at(p - j)) * w[r + j];
    out[i] = tmp;
}

// sm = smoothed / (bleed + eps)   (bleed: [H][W], shared by the batch)
__global__ void om_norm_kernel(double* __restrict__ sm, const double* __restrict__ bleed, int64_t total, int64_t hw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    sm[i] = sm[i] / (bleed[i % hw] + 2.220446049250313e-16);
}

// scipy.ndimage.sobel building blocks, mode 'reflect' (x[-1] = x[0], x[n] = x[n - 1]):
// deriv != 0: correlate1d with [-1, 0, 1] (antisymmetric: tmp = x[i] * 0 + (x[i - 1] - x[i + 1]) * -1)
// else      : correlate1d with [ 1, 2, 1] (symmetric:     tmp = x[i] * 2 + (x[i - 1] + x[i + 1]) * 1)
__global__ void om_sobel_kernel(const double* __restrict__ in, double* __restrict__ out, int B, int H, int W, int axis, int deriv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)B * H * W;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int p = axis ? x : y, n = axis ? W : H;
    const int64_t step = axis ? 1 : W;
    const double c = in[i];
    const double lo = in[i + (int64_t)((p > 0 ? p - 1 : 0) - p) * step];
    const double hi = in[i + (int64_t)((p < n - 1 ? p + 1 : n - 1) - p) * step];
    double tmp;
    if (deriv) {
        tmp = c * 0.0;
        tmp += (lo - hi) * -1.0;
    } else {
        tmp = c * 2.0;
        tmp += (lo + hi) * 1.0;
    }
    out[i] = tmp;
}

// np.hypot on this image's libm (glibc 2.35, sysdeps/ieee754/dbl-64/e_hypot.c, the path without a fast FMA): h = sqrt(ax^2 + ay^2)
// followed by one correction step; 9 M random inputs agree with numpy bit for bit (tests/test_data_cpu.py pins the restatement).
__device__ __forceinline__ double hypot64(double x, double y) {
    x = fabs(x);
    y = fabs(y);
    double ax = x < y ? y : x, ay = x < y ? x : y;
    double scale = 1.0;
    if (ax > 0x1p+511) {
        if (ay <= ax * 0x1p-54) return ax + ay;
        ax *= 0x1p-600; ay *= 0x1p-600; scale = 0x1p+600;
    } else if (ay < 0x1p-459) {
        if (ax >= ay / 0x1p-54) return ax + ay;
        ax *= 0x1p+600; ay *= 0x1p+600; scale = 0x1p-600;
    } else if (ax >= ay / 0x1p-54) {
        return ax + ay;
    }
    double h = sqrt(ax * ax + ay * ay);
    double t1, t2;
    if (h <= 2.0 * ay) {
        const double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        const double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h * scale;
}

__global__ void om_mag_kernel(const double* __restrict__ is, const double* __restrict__ js, double* __restrict__ mag, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    mag[i] = hypot64(is[i], js[i]);
}

// non-maximum suppression + low threshold (dataset_generator._canny, expression by expression); border pixels are 0
__global__ void om_nms_kernel(const double* __restrict__ is, const double* __restrict__ js, const double* __restrict__ mag,
                              double* __restrict__ out, int B, int H, int W, double low) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)B * H * W;
    if (idx >= total) return;
    const int x = (int)(idx % W), y = (int)((idx / W) % H);
    if (H < 3 || W < 3 || x == 0 || y == 0 || x == W - 1 || y == H - 1) { out[idx] = 0.0; return; }
    const double i = is[idx], j = js[idx], m = mag[idx];
    const double ai = fabs(i), aj = fabs(j);
    auto sh = [&](int dx, int dy) -> double { return mag[idx + (int64_t)dx * W + dy]; };      // (row + dx, col + dy)
    const bool up = i >= 0, down = i <= 0, left = j <= 0, right = j >= 0;
    const bool c1 = (up && right) || (down && left);
    const bool c2 = !c1 && ((down && right) || (up && left));
    const double w_ji = ai > 0 ? aj / ai : 0.0, w_ij = aj > 0 ? ai / aj : 0.0;
    bool keep = false;
    auto test = [&](double wgt, double n11, double n12, double n21, double n22) {
        const bool plus = (n12 * wgt + n11 * (1.0 - wgt)) <= m;
        const bool minus = (n22 * wgt + n21 * (1.0 - wgt)) <= m;
        if (plus && minus) keep = true;
    };
    if (c1) {
        if (ai > aj) test(w_ji, sh(1, 0), sh(1, 1), sh(-1, 0), sh(-1, -1));
        else test(w_ij, sh(0, 1), sh(1, 1), sh(0, -1), sh(-1, -1));
    } else if (c2) {
        if (ai < aj) test(w_ij, sh(0, 1), sh(-1, 1), sh(0, -1), sh(1, -1));
        else test(w_ji, sh(-1, 0), sh(-1, 1), sh(1, 0), sh(1, -1));
    }
    out[idx] = (keep && m >= low) ? m : 0.0;
}

// ---- per-image iterative passes: one workgroup of 1024 threads per image, planes in global memory (L2-resident) ----
constexpr int OMT = 1024;

// seeds = thin >= high (and > 0), grown through the 8-neighbourhood inside low = thin > 0 until nothing changes: the pixels of the
// 8-connected components of the low mask that hold a high pixel (skimage's hysteresis by labelling, as a fixed point)
__global__ __launch_bounds__(OMT) void om_hysteresis_kernel(const double* __restrict__ thin, uint8_t* __restrict__ edges, int H, int W,
                                                            double high) {
    const int64_t base = (int64_t)blockIdx.x * H * W;
    const double* t = thin + base;
    uint8_t* e = edges + base;
    const int n = H * W;
    for (int p = threadIdx.x; p < n; p += OMT) e[p] = (t[p] > 0 && t[p] >= high) ? 1 : 0;
    __syncthreads();
    for (;;) {
        int changed = 0;
        for (int p = threadIdx.x; p < n; p += OMT) {
            if (e[p] || !(t[p] > 0)) continue;
            const int y = p / W, x = p - y * W;
            bool hit = false;
            for (int dy = -1; dy <= 1 && !hit; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && e[yy * W + xx]) { hit = true; break; }
                }
            if (hit) { e[p] = 1; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }
}

// binary dilation (any) / erosion (all) with a k x k square whose centre is at index k / 2 (scipy.ndimage, origin 0), outside the
// image = 0 (border_value 0): offsets -(k / 2) .. k - 1 - k / 2 on both axes
__global__ void om_morph_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H, int W, int k, int erode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)B * H * W;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int lo = -(k / 2), hi = k - 1 - k / 2;
    bool any = false, all = true;
    for (int dy = lo; dy <= hi; ++dy)
        for (int dx = lo; dx <= hi; ++dx) {
            const int yy = y + dy, xx = x + dx;
            const bool v = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && in[i + (int64_t)dy * W + dx];
            any = any || v;
            all = all && v;
        }
    out[i] = erode ? (uint8_t)all : (uint8_t)any;
}

// scipy.ndimage.binary_fill_holes(m, 3 x 3): the background reachable from outside the image through the 8-neighbourhood stays
// background, everything else becomes foreground.  out = !(outside-connected background); `bg` is scratch.
__global__ __launch_bounds__(OMT) void om_fill_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ bg, uint8_t* __restrict__ out,
                                                      int H, int W) {
    const int64_t base = (int64_t)blockIdx.x * H * W;
    const uint8_t* m = in + base;
    uint8_t* g = bg + base;
    const int n = H * W;
    for (int p = threadIdx.x; p < n; p += OMT) {
        const int y = p / W, x = p - y * W;
        g[p] = (!m[p] && (y == 0 || x == 0 || y == H - 1 || x == W - 1)) ? 1 : 0;      // background on the frame touches the outside
    }
    __syncthreads();
    for (;;) {
        int changed = 0;
        for (int p = threadIdx.x; p < n; p += OMT) {
            if (g[p] || m[p]) continue;
            const int y = p / W, x = p - y * W;
            bool hit = false;
            for (int dy = -1; dy <= 1 && !hit; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && g[yy * W + xx]) { hit = true; break; }
                }
            if (hit) { g[p] = 1; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }
    for (int p = threadIdx.x; p < n; p += OMT) out[base + p] = g[p] ? 0 : 1;
}

// Largest 8-connected component of m (ties: the one whose first pixel comes first in raster order; no component at all: every
// pixel, as `labels == argmax(bincount)` gives for an empty mask).  lab / cnt: int32 scratch planes.
__global__ __launch_bounds__(OMT) void om_largest_kernel(const uint8_t* __restrict__ in, int* __restrict__ lab, int* __restrict__ cnt,
                                                         uint8_t* __restrict__ out, int H, int W) {
    const int64_t base = (int64_t)blockIdx.x * H * W;
    const uint8_t* m = in + base;
    int* L = lab + base;
    int* C = cnt + base;
    const int n = H * W;
    for (int p = threadIdx.x; p < n; p += OMT) { L[p] = m[p] ? p : n; C[p] = 0; }
    __syncthreads();
    for (;;) {                              // label = smallest pixel index of the component: neighbour minimum + pointer jumping
        int changed = 0;
        for (int p = threadIdx.x; p < n; p += OMT) {
            if (!m[p]) continue;
            const int y = p / W, x = p - y * W;
            int best = L[p];
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && m[yy * W + xx]) {
                        const int v = L[yy * W + xx];
                        best = v < best ? v : best;
                    }
                }
            const int j = L[best];          // the label's own label (labels are pixel indices of the same component)
            best = j < best ? j : best;
            if (best < L[p]) { L[p] = best; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }
    for (int p = threadIdx.x; p < n; p += OMT)
        if (m[p]) atomicAdd(&C[L[p]], 1);
    __syncthreads();
    __shared__ int s_cnt[OMT], s_lab[OMT];
    int bc = 0, bl = n;
    for (int p = threadIdx.x; p < n; p += OMT) {
        const int c = C[p];
        if (c > bc || (c == bc && c > 0 && p < bl)) { bc = c; bl = p; }
    }
    s_cnt[threadIdx.x] = bc;
    s_lab[threadIdx.x] = bl;
    __syncthreads();
    for (int o = OMT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const int c2 = s_cnt[threadIdx.x + o], l2 = s_lab[threadIdx.x + o];
            if (c2 > s_cnt[threadIdx.x] || (c2 == s_cnt[threadIdx.x] && l2 < s_lab[threadIdx.x])) { s_cnt[threadIdx.x] = c2; s_lab[threadIdx.x] = l2; }
        }
        __syncthreads();
    }
    const int win = s_cnt[0] > 0 ? s_lab[0] : -1;
    for (int p = threadIdx.x; p < n; p += OMT) out[base + p] = win < 0 ? 1 : (uint8_t)(m[p] && L[p] == win);
}

inline unsigned g1(int64_t n) { return (unsigned)cdiv64(n, 256); }

}  // namespace

// bytes of scratch ssad_obj_mask needs for a batch
extern "C" int64_t ssad_obj_mask_workspace(int B, int H, int W) {
    const int64_t px = (int64_t)B * H * W;
    return px * (4 * 8 + 2 * 4 + 3) + (int64_t)H * W * 16 + 64 * 8 + 256;
}

extern "C" int ssad_obj_mask(const uint8_t* rgb, uint8_t* mask, uint8_t* edges, int B, int H, int W, const double* gauss_w_host,
                             int radius, double low, double high, void* workspace, void* stream) {
    SSAD_CHECK_ARG(rgb && mask && edges && workspace && gauss_w_host && B > 0 && H > 0 && W > 0, "bad argument");
    SSAD_CHECK_ARG(radius >= 0 && radius <= 31, "Gaussian radius up to 31");
    SSAD_CHECK_ARG((int64_t)H * W < (int64_t)1 << 30, "image too large");
    hipStream_t st = (hipStream_t)stream;
    const int64_t px = (int64_t)B * H * W, hw = (int64_t)H * W;
    char* ws = (char*)workspace;
    double* b0 = (double*)ws;            ws += px * 8;
    double* b1 = (double*)ws;            ws += px * 8;
    double* b2 = (double*)ws;            ws += px * 8;
    double* b3 = (double*)ws;            ws += px * 8;
    double* bleed = (double*)ws;         ws += hw * 8;
    double* bleed_t = (double*)ws;       ws += hw * 8;
    double* wdev = (double*)ws;          ws += 64 * 8;
    int* lab = (int*)ws;                 ws += px * 4;
    int* cnt = (int*)ws;                 ws += px * 4;
    uint8_t* m0 = (uint8_t*)ws;          ws += px;
    uint8_t* m1 = (uint8_t*)ws;          ws += px;
    uint8_t* m2 = (uint8_t*)ws;
    if (hipMemcpyAsync(wdev, gauss_w_host, (size_t)(2 * radius + 1) * 8, hipMemcpyHostToDevice, st) != hipSuccess) {
        ssad_set_error("ssad_obj_mask: weight upload failed");
        return 1;
    }
    const dim3 blk(256);
    // bleed = gaussian(ones) along rows, then columns (scipy filters axis 0 first)
    hipLaunchKernelGGL(om_gauss_kernel, dim3(g1(hw)), blk, 0, st, (const double*)nullptr, bleed_t, 1, H, W, 0, wdev, radius, 1);
    hipLaunchKernelGGL(om_gauss_kernel, dim3(g1(hw)), blk, 0, st, bleed_t, bleed, 1, H, W, 1, wdev, radius, 0);
    hipLaunchKernelGGL(om_gray_kernel, dim3(g1(px)), blk, 0, st, rgb, b0, px);
    hipLaunchKernelGGL(om_gauss_kernel, dim3(g1(px)), blk, 0, st, b0, b1, B, H, W, 0, wdev, radius, 0);
    hipLaunchKernelGGL(om_gauss_kernel, dim3(g1(px)), blk, 0, st, b1, b0, B, H, W, 1, wdev, radius, 0);
    hipLaunchKernelGGL(om_norm_kernel, dim3(g1(px)), blk, 0, st, b0, bleed, px, hw);                  // b0 = smoothed
    hipLaunchKernelGGL(om_sobel_kernel, dim3(g1(px)), blk, 0, st, b0, b1, B, H, W, 1, 1);             // d / d column ...
    hipLaunchKernelGGL(om_sobel_kernel, dim3(g1(px)), blk, 0, st, b1, b2, B, H, W, 0, 0);             // ... smoothed along rows: jsobel
    hipLaunchKernelGGL(om_sobel_kernel, dim3(g1(px)), blk, 0, st, b0, b1, B, H, W, 0, 1);             // d / d row ...
    hipLaunchKernelGGL(om_sobel_kernel, dim3(g1(px)), blk, 0, st, b1, b3, B, H, W, 1, 0);             // ... smoothed along columns: isobel
    hipLaunchKernelGGL(om_mag_kernel, dim3(g1(px)), blk, 0, st, b3, b2, b1, px);                      // b1 = magnitude
    hipLaunchKernelGGL(om_nms_kernel, dim3(g1(px)), blk, 0, st, b3, b2, b1, b0, B, H, W, low);        // b0 = thinned magnitude
    hipLaunchKernelGGL(om_hysteresis_kernel, dim3(B), dim3(OMT), 0, st, b0, edges, H, W, high);
    hipLaunchKernelGGL(om_morph_kernel, dim3(g1(px)), blk, 0, st, edges, m0, B, H, W, 3, 0);          // binary_dilation(edges, 3 x 3)
    hipLaunchKernelGGL(om_morph_kernel, dim3(g1(px)), blk, 0, st, m0, m1, B, H, W, 3, 0);             // binary_closing = dilation ...
    hipLaunchKernelGGL(om_morph_kernel, dim3(g1(px)), blk, 0, st, m1, m0, B, H, W, 3, 1);             // ... then erosion
    hipLaunchKernelGGL(om_fill_kernel, dim3(B), dim3(OMT), 0, st, m0, m2, m1, H, W);                  // m1 = holes filled
    hipLaunchKernelGGL(om_morph_kernel, dim3(g1(px)), blk, 0, st, m1, m0, B, H, W, 4, 1);             // binary_erosion(., 4 x 4)
    hipLaunchKernelGGL(om_largest_kernel, dim3(B), dim3(OMT), 0, st, m0, lab, cnt, mask, H, W);
    SSAD_CHECK_LAUNCH();
    return 0;
}
