// Batched synthetic-defect augmentation on the GPU: RandomAffine / crop, polygon cut-paste, rotated scars, poly-lines,
// ColorJitter, ToTensor + Normalize -- one uint8 HWC batch in, one fp32 NCHW batch out, BYTE-IDENTICAL to the PIL path.
//
// Replaces the pixel work of PretextTaskDataset.__getitem__ (src/self_supervised/datasets.py:209-394) and of
// dataset_generator.rect2poly / paste_patch (src/self_supervised/dataset_generator.py:42-101, :268-275), which the
// reference runs with PIL inside 8 DataLoader worker processes.  The random *parameters* (label, boxes, polygon
// vertices, angles, jitter factors) are drawn on the host in the reference's order (augment.py) and arrive as one
// ssad_aug_params record per sample; everything per-pixel happens here, by Pillow's own rules (restated in
// self_supervised/pil_exact.py and pinned against Pillow there):
//   * Image.transform(AFFINE, NEAREST) / Image.rotate(expand=True): 16.16 fixed-point source coordinates (affine_fixed);
//   * ImageDraw.polygon: scan-line intersections in float32, doubled lower end points, ROUND_UP / ROUND_DOWN spans;
//   * ImageDraw.line: Bresenham for width 1, one polygon quadrilateral per segment for wider lines;
//   * ImageEnhance: Image.blend in float32 (separate multiply and add, truncation, clamping only when extrapolating),
//     integer luma (19595, 38470, 7471), the Contrast mean taken from the image as it is when Contrast runs.
// HBM-bound byte work: one thread per output pixel (or per line segment), no LDS needed.
#include "common.h"
#include "../../include/ssad.h"

// Pillow's C code rounds every float32 product and sum separately (x86-64 has no fused multiply-add in its baseline):
// hipcc's default -ffp-contract=fast would fuse `a * b + c` below into one v_fma_f32 and move scan-line intersections /
// blend results across a rounding boundary (seen as isolated pixels on long poly-lines).  No contraction in this file.
#pragma clang fp contract(off)

namespace {

// (int)in1 + alpha * ((int)in2 - (int)in1) in float32 with separate roundings (no fma contraction), Blend.c
__device__ __forceinline__ int blend_u8(int degenerate, int v, float alpha) {
    const float t = __fadd_rn((float)degenerate, __fmul_rn(alpha, (float)(v - degenerate)));
    if (alpha >= 0.f && alpha <= 1.f) return (int)t;                  // interpolation: plain truncation
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__device__ __forceinline__ int round_up(float f) {
    return f >= 0.f ? (int)floorf(__fadd_rn(f, 0.5f)) : -(int)floorf(__fadd_rn(fabsf(f), 0.5f));
}
__device__ __forceinline__ int round_down(float f) {
    return f >= 0.f ? (int)ceilf(__fsub_rn(f, 0.5f)) : -(int)ceilf(__fsub_rn(fabsf(f), 0.5f));
}

// Does ImageDraw.polygon(vertices, fill) paint pixel (px, py) of a W x H image?  (pil_exact.polygon_row_spans)
__device__ bool polygon_covers(const int32_t* __restrict__ xy, int n, int W, int H, int px, int py) {
    if ((unsigned)px >= (unsigned)W || (unsigned)py >= (unsigned)H) return false;
    int pymin = H - 1, pymax = 0;
    bool hit = false;
    float xx[16];
    for (int i = 0; i < n; ++i) {
        const int y0 = xy[2 * i + 1], y1 = xy[2 * ((i + 1) % n) + 1];
        pymin = min(pymin, min(y0, y1));
        pymax = max(pymax, max(y0, y1));
    }
    pymin = max(pymin, 0);
    pymax = min(pymax, H);
    int j = 0;
    for (int i = 0; i < n; ++i) {
        const int x0 = xy[2 * i], y0 = xy[2 * i + 1], x1 = xy[2 * ((i + 1) % n)], y1 = xy[2 * ((i + 1) % n) + 1];
        if (y0 == y1) {                                   // horizontal edges are drawn as they are
            if (y0 == py && px >= min(x0, x1) && px <= max(x0, x1)) hit = true;
            continue;
        }
        const int ymin = min(y0, y1), ymax = max(y0, y1);
        if (py < pymin || py > pymax || py < ymin || py > ymax) continue;
        const float dx = __fdiv_rn((float)(x1 - x0), (float)(y1 - y0));
        const float v = __fadd_rn(__fmul_rn((float)(py - y0), dx), (float)x0);
        xx[j++] = v;
        if (py == ymax && py < pymax) xx[j++] = v;        // "needed to draw consistent polygons"
    }
    if (hit) return true;
    for (int a = 1; a < j; ++a) {                         // insertion sort (j <= 16)
        const float key = xx[a];
        int b = a - 1;
        while (b >= 0 && xx[b] > key) { xx[b + 1] = xx[b]; --b; }
        xx[b + 1] = key;
    }
    int x_pos = j ? (int)xx[0] : 0;
    for (int i = 1; i < j; i += 2) {
        const int x_end = round_down(xx[i]);
        if (x_end < x_pos) continue;
        int x_start = round_up(xx[i - 1]);
        if (x_pos > x_start) {
            x_start = x_pos;
            if (x_end < x_start) continue;
        }
        if (px >= x_start && px <= x_end) return true;
        x_pos = x_end + 1;
    }
    return false;
}

__device__ __forceinline__ void fetch(const uint8_t* img, int H, int W, int y, int x, int* rgb) {
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
        const uint8_t* p = img + ((int64_t)y * W + x) * 3;
        rgb[0] = p[0]; rgb[1] = p[1]; rgb[2] = p[2];
    } else {
        rgb[0] = rgb[1] = rgb[2] = 0;
    }
}

// a pixel of the defect source: a flat colour, or the crop [src_left + lx, src_top + ly] of the cutting window (Image.crop
// pads with zeros outside the window), then the optional twofold Brightness enhancement
__device__ __forceinline__ void source_pixel(const uint8_t* cut, const ssad_aug_params& p, int H, int W, int flat, const int32_t* frgb,
                                             int src_left, int src_top, int lx, int ly, int nbright, const float* bright, int* s) {
    if (flat) {
        s[0] = frgb[0]; s[1] = frgb[1]; s[2] = frgb[2];
    } else {
        const int cx = src_left + lx, cy = src_top + ly;
        if ((unsigned)cx < (unsigned)p.cut_w && (unsigned)cy < (unsigned)p.cut_h) fetch(cut, H, W, p.cut_top + cy, p.cut_left + cx, s);
        else s[0] = s[1] = s[2] = 0;
    }
    for (int k = 0; k < nbright; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) s[c] = blend_u8(0, s[c], bright[k]);
}

// stage 1: affine / crop, polygon patch or scars -> uint8 HWC work image of size h x w
__global__ void compose_kernel(const uint8_t* __restrict__ imgs, const uint8_t* __restrict__ cuts,
                               const ssad_aug_params* __restrict__ params, uint8_t* __restrict__ work, int B, int H, int W,
                               int h, int w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * h * w) return;
    const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((int64_t)w * h));
    const ssad_aug_params& p = params[b];
    const uint8_t* img = imgs + (int64_t)b * H * W * 3;
    const uint8_t* cut = p.cut_index >= 0 ? cuts + (int64_t)p.cut_index * H * W * 3 : img;
    int rgb[3];
    {
        const int ox = x + p.crop_left, oy = y + p.crop_top;
        if (p.aff_on) {
            const int sx = (p.aff_fix[2] + oy * p.aff_fix[1] + ox * p.aff_fix[0]) >> 16;
            const int sy = (p.aff_fix[5] + oy * p.aff_fix[4] + ox * p.aff_fix[3]) >> 16;
            fetch(img, H, W, sy, sx, rgb);
        } else {
            fetch(img, H, W, oy, ox, rgb);
        }
    }
    if (p.label == 1 && p.patch_w > 0) {
        const int lx = x - p.patch_dst_left, ly = y - p.patch_dst_top;
        if ((unsigned)lx < (unsigned)p.patch_w && (unsigned)ly < (unsigned)p.patch_h &&
            polygon_covers(p.poly_xy, p.poly_n, p.patch_w, p.patch_h, lx, ly))
            source_pixel(cut, p, H, W, p.patch_flat, p.patch_rgb, p.patch_src_left, p.patch_src_top, lx, ly, p.patch_nbright,
                         p.patch_bright, rgb);
    } else if (p.label == 2) {
        for (int k = 0; k < p.scar_n; ++k) {              // pasted in order: a later copy overwrites an earlier one
            const int lx = x - p.scar_dst[2 * k], ly = y - p.scar_dst[2 * k + 1];
            if ((unsigned)lx >= (unsigned)p.scar_rw || (unsigned)ly >= (unsigned)p.scar_rh) continue;
            int sx = lx, sy = ly;
            if (p.scar_rot) {                             // Image.rotate(angle, expand=True), NEAREST, transparent outside
                sx = (p.scar_fix[2] + ly * p.scar_fix[1] + lx * p.scar_fix[0]) >> 16;
                sy = (p.scar_fix[5] + ly * p.scar_fix[4] + lx * p.scar_fix[3]) >> 16;
            }
            if ((unsigned)sx >= (unsigned)p.scar_w || (unsigned)sy >= (unsigned)p.scar_h) continue;
            source_pixel(cut, p, H, W, p.scar_flat, p.scar_rgb, p.scar_src_left, p.scar_src_top, sx, sy, p.scar_nbright,
                         p.scar_bright, rgb);
        }
    }
    uint8_t* o = work + i * 3;
    o[0] = (uint8_t)rgb[0]; o[1] = (uint8_t)rgb[1]; o[2] = (uint8_t)rgb[2];
}

// stage 1b: poly-lines, one thread per (sample, segment).  Every writer stores the same colour, so overlapping segments
// need no ordering.
__device__ __forceinline__ void put(uint8_t* img, int h, int w, int x, int y, const int32_t* rgb) {
    if ((unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h) {
        uint8_t* o = img + ((int64_t)y * w + x) * 3;
        o[0] = (uint8_t)rgb[0]; o[1] = (uint8_t)rgb[1]; o[2] = (uint8_t)rgb[2];
    }
}

__global__ void line_kernel(const ssad_aug_params* __restrict__ params, uint8_t* __restrict__ work, int B, int h, int w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = t / (SSAD_AUG_MAX_LINE_POINTS - 1), seg = t % (SSAD_AUG_MAX_LINE_POINTS - 1);
    if (b >= B) return;
    const ssad_aug_params& p = params[b];
    if (p.label != 3 || seg + 1 >= p.line_n) return;
    uint8_t* img = work + (int64_t)b * h * w * 3;
    int x0 = p.line_xy[2 * seg], y0 = p.line_xy[2 * seg + 1];
    const int x1 = p.line_xy[2 * seg + 2], y1 = p.line_xy[2 * seg + 3];
    if (p.line_width <= 1) {                              // Draw.c line8 / line32: all pixels but the end point ...
        int dx = x1 - x0, xs = 1, dy = y1 - y0, ys = 1;
        if (dx < 0) { dx = -dx; xs = -1; }
        if (dy < 0) { dy = -dy; ys = -1; }
        if (dx == 0) {
            for (int i = 0; i < dy; ++i) { put(img, h, w, x0, y0, p.line_rgb); y0 += ys; }
        } else if (dy == 0) {
            for (int i = 0; i < dx; ++i) { put(img, h, w, x0, y0, p.line_rgb); x0 += xs; }
        } else if (dx > dy) {
            const int n = dx;
            dy += dy;
            int e = dy - dx;
            dx += dx;
            for (int i = 0; i < n; ++i) {
                put(img, h, w, x0, y0, p.line_rgb);
                if (e >= 0) { y0 += ys; e -= dx; }
                e += dy;
                x0 += xs;
            }
        } else {
            const int n = dy;
            dx += dx;
            int e = dx - dy;
            dy += dy;
            for (int i = 0; i < n; ++i) {
                put(img, h, w, x0, y0, p.line_rgb);
                if (e >= 0) { x0 += xs; e -= dy; }
                e += dx;
                y0 += ys;
            }
        }
        if (seg + 2 == p.line_n) put(img, h, w, x1, y1, p.line_rgb);        // ... then ImagingDrawPoint on the last point
        return;
    }
    if (!p.line_quad_ok[seg]) {                           // ImagingDrawWideLine on a zero-length segment: one point
        put(img, h, w, x0, y0, p.line_rgb);
        return;
    }
    const int32_t* q = p.line_quad + 8 * seg;
    int bx0 = q[0], bx1 = q[0], by0 = q[1], by1 = q[1];
    for (int k = 1; k < 4; ++k) {
        bx0 = min(bx0, q[2 * k]); bx1 = max(bx1, q[2 * k]);
        by0 = min(by0, q[2 * k + 1]); by1 = max(by1, q[2 * k + 1]);
    }
    bx0 = max(bx0, 0); by0 = max(by0, 0); bx1 = min(bx1, w - 1); by1 = min(by1, h - 1);
    for (int yy = by0; yy <= by1; ++yy)
        for (int xq = bx0; xq <= bx1; ++xq)
            if (polygon_covers(q, 4, w, h, xq, yy)) put(img, h, w, xq, yy, p.line_rgb);
}

// ColorJitter op k of the sampled order on one pixel; `mean` = the Contrast degenerate level
__device__ __forceinline__ void jitter_op(int op, float f, int mean, int& r, int& g, int& b) {
    if (op == 0) {
        r = blend_u8(0, r, f); g = blend_u8(0, g, f); b = blend_u8(0, b, f);
    } else if (op == 1) {
        r = blend_u8(mean, r, f); g = blend_u8(mean, g, f); b = blend_u8(mean, b, f);
    } else {
        const int l = luma(r, g, b);
        r = blend_u8(l, r, f); g = blend_u8(l, g, f); b = blend_u8(l, b, f);
    }
}

// stage 2: ImageEnhance.Contrast's degenerate level = int(mean of the L channel + 0.5) of the image AS IT IS when Contrast
// runs, i.e. after the jitter ops sampled before it (re-applied per pixel inside the reduction)
__global__ void gray_mean_kernel(const uint8_t* __restrict__ work, const ssad_aug_params* __restrict__ params,
                                 float* __restrict__ mean, int hw) {
    __shared__ unsigned long long sh[256];
    const ssad_aug_params& p = params[blockIdx.x];
    const uint8_t* px = work + (int64_t)blockIdx.x * hw * 3;
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) {
        int r = px[3 * i], g = px[3 * i + 1], b = px[3 * i + 2];
        for (int k = 0; k < p.jit_n && p.jit_order[k] != 1; ++k) jitter_op(p.jit_order[k], p.jit_factor[p.jit_order[k]], 0, r, g, b);
        s += (unsigned long long)luma(r, g, b);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[blockIdx.x] = (float)(int)((double)sh[0] / (double)hw + 0.5);
}

// stage 3: colour jitter in the sampled order, then ToTensor + Normalize -> NCHW fp32
__global__ void jitter_normalize_kernel(const uint8_t* __restrict__ work, const ssad_aug_params* __restrict__ params,
                                        const float* __restrict__ gmean, float* __restrict__ out, int B, int h, int w,
                                        float m0, float m1, float m2, float s0, float s1, float s2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t hw = (int64_t)h * w;
    if (i >= (int64_t)B * hw) return;
    const int b = (int)(i / hw);
    const int64_t pix = i - (int64_t)b * hw;
    const ssad_aug_params& p = params[b];
    int r = work[i * 3], g = work[i * 3 + 1], bl = work[i * 3 + 2];
    const int mean = (int)gmean[b];
    for (int k = 0; k < p.jit_n; ++k) jitter_op(p.jit_order[k], p.jit_factor[p.jit_order[k]], mean, r, g, bl);
    // ToTensor (u8 / 255) then Normalize ((t - mean) / std): IEEE single operations, as torch evaluates them
    out[((int64_t)b * 3 + 0) * hw + pix] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)r, 255.f), m0), s0);
    out[((int64_t)b * 3 + 1) * hw + pix] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)g, 255.f), m1), s1);
    out[((int64_t)b * 3 + 2) * hw + pix] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)bl, 255.f), m2), s2);
}

// uint8 HWC -> fp32 CHW in [0,1] (the "original" the Dataset returns as third element)
__global__ void u8_to_f32_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int B, int64_t hw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * hw) return;
    const int b = (int)(i / hw);
    const int64_t pix = i - (int64_t)b * hw;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((int64_t)b * 3 + c) * hw + pix] = __fdiv_rn((float)img[i * 3 + c], 255.f);
}

// uint8 HWC -> both outputs of MVTecDataset.__getitem__ in one pass: ToTensor (orig, may be NULL) and Normalize(ToTensor) (norm)
__global__ void u8_to_f32_norm_kernel(const uint8_t* __restrict__ img, float* __restrict__ orig, float* __restrict__ norm, int B,
                                      int64_t hw, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * hw) return;
    const int b = (int)(i / hw);
    const int64_t pix = i - (int64_t)b * hw;
    const float m[3] = {m0, m1, m2}, s[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t = __fdiv_rn((float)img[i * 3 + c], 255.f);
        if (orig) orig[((int64_t)b * 3 + c) * hw + pix] = t;
        norm[((int64_t)b * 3 + c) * hw + pix] = __fdiv_rn(__fsub_rn(t, m[c]), s[c]);
    }
}

}  // namespace

extern "C" int ssad_aug_params_size(void) { return (int)sizeof(ssad_aug_params); }

extern "C" int ssad_cutpaste_augment(const uint8_t* imgs, const uint8_t* cuts, const ssad_aug_params* params, uint8_t* work,
                                     float* gray_mean, float* out, int B, int H, int W, int h, int w, const float* mean3_host,
                                     const float* std3_host, void* stream) {
    SSAD_CHECK_ARG(imgs && params && work && gray_mean && out && mean3_host && std3_host, "null pointer");
    SSAD_CHECK_ARG(B > 0 && H > 0 && W > 0 && h > 0 && w > 0, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)B * h * w;
    hipLaunchKernelGGL(compose_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, imgs, cuts ? cuts : imgs, params,
                       work, B, H, W, h, w);
    hipLaunchKernelGGL(line_kernel, dim3((unsigned)cdiv64((int64_t)B * (SSAD_AUG_MAX_LINE_POINTS - 1), 64)), dim3(64), 0, st, params,
                       work, B, h, w);
    hipLaunchKernelGGL(gray_mean_kernel, dim3(B), dim3(256), 0, st, work, params, gray_mean, h * w);
    hipLaunchKernelGGL(jitter_normalize_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, work, params, gray_mean, out, B,
                       h, w, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    SSAD_CHECK_LAUNCH();
    return 0;
}

// HOST helper of the sampler (augment.sample_defect): channel sums of the window [top, top + h) x [left, left + w) of
// Image.transform((W, H), AFFINE, NEAREST) of an H x W x 3 uint8 image (fix = Pillow's 16.16 coefficients; NULL = the image
// itself), zero outside the image -- what np.array(img.transform(...).crop(...)).sum((0, 1)) gives, in integers (exact).  The
// colour-similarity test of datasets.py:300-312 needs this mean BEFORE the next random draw, so it cannot wait for the GPU.
extern "C" int ssad_affine_window_sum_u8(const uint8_t* img, int H, int W, const int32_t* fix, int left, int top, int w, int h,
                                         int64_t* sum3) {
    SSAD_CHECK_ARG(img && sum3 && H > 0 && W > 0 && w > 0 && h > 0 && left >= 0 && top >= 0, "bad argument");
    int64_t s0 = 0, s1 = 0, s2 = 0;
    const int y1 = top + h < H ? top + h : H, x1 = left + w < W ? left + w : W;
    for (int y = top; y < y1; ++y) {
        if (!fix) {
            const uint8_t* row = img + ((int64_t)y * W + left) * 3;
            for (int x = left; x < x1; ++x, row += 3) { s0 += row[0]; s1 += row[1]; s2 += row[2]; }
            continue;
        }
        int64_t fx = (int64_t)fix[2] + (int64_t)y * fix[1] + (int64_t)left * fix[0];
        int64_t fy = (int64_t)fix[5] + (int64_t)y * fix[4] + (int64_t)left * fix[3];
        for (int x = left; x < x1; ++x, fx += fix[0], fy += fix[3]) {
            const int64_t sx = fx >> 16, sy = fy >> 16;         // arithmetic shift = floor, as Pillow's affine_fixed
            if (sx >= 0 && sx < W && sy >= 0 && sy < H) {
                const uint8_t* px = img + (sy * W + sx) * 3;
                s0 += px[0]; s1 += px[1]; s2 += px[2];
            }
        }
    }
    sum3[0] = s0; sum3[1] = s1; sum3[2] = s2;
    return 0;
}

extern "C" int ssad_u8hwc_to_f32chw(const uint8_t* img, float* out, int B, int H, int W, void* stream) {
    SSAD_CHECK_ARG(img && out && B > 0 && H > 0 && W > 0, "bad argument");
    const int64_t total = (int64_t)B * H * W;
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, img, out, B,
                       (int64_t)H * W);
    SSAD_CHECK_LAUNCH();
    return 0;
}

extern "C" int ssad_u8hwc_to_f32chw_norm(const uint8_t* img, float* orig, float* norm, int B, int H, int W, const float* mean3_host,
                                         const float* std3_host, void* stream) {
    SSAD_CHECK_ARG(img && norm && mean3_host && std3_host && B > 0 && H > 0 && W > 0, "bad argument");
    const int64_t total = (int64_t)B * H * W;
    hipLaunchKernelGGL(u8_to_f32_norm_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, img, orig, norm,
                       B, (int64_t)H * W, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    SSAD_CHECK_LAUNCH();
    return 0;
}
