// Batched synthetic-defect augmentation on the GPU: crop / affine, polygon cut-paste, rotated scars, poly-lines,
// colour jitter, ToTensor + Normalize -- one uint8 HWC batch in, one fp32 NCHW batch out.
//
// Replaces the pixel work of PretextTaskDataset.__getitem__ (src/self_supervised/datasets.py:209-394) and of
// dataset_generator.rect2poly / paste_patch (src/self_supervised/dataset_generator.py:42-101, :268-275), which the
// reference runs with PIL inside 8 DataLoader worker processes.  The random *parameters* (label, boxes, polygon
// vertices, angles, jitter factors) are drawn on the host in the reference's order (augment.py) and arrive as one
// ssad_aug_params record per sample; everything per-pixel happens here.  HBM-bound byte work: one thread per
// output pixel, three channels per thread, no LDS needed.
#include "common.h"
#include "../../include/ssad.h"

namespace {

__device__ __forceinline__ float clamp255(float v) { return v <= 0.f ? 0.f : (v >= 255.f ? 255.f : floorf(v)); }

// PIL ImageEnhance = Image.blend(degenerate, image, factor) with uint8 truncation
__device__ __forceinline__ float blend_u8(float degenerate, float v, float f) {
    return clamp255(degenerate + f * (v - degenerate));
}

__device__ __forceinline__ float gray_u8(float r, float g, float b) {
    // PIL "L": (R*19595 + G*38470 + B*7471 + 0x8000) >> 16
    return floorf((r * 19595.f + g * 38470.f + b * 7471.f + 32768.f) / 65536.f);
}

__device__ bool in_polygon(const float* xy, int n, float px, float py) {
    bool in = false;
    for (int i = 0, j = n - 1; i < n; j = i++) {
        const float xi = xy[2 * i], yi = xy[2 * i + 1], xj = xy[2 * j], yj = xy[2 * j + 1];
        if (((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi)) in = !in;
    }
    return in;
}

__device__ float seg_dist2(float px, float py, float ax, float ay, float bx, float by) {
    const float vx = bx - ax, vy = by - ay, wx = px - ax, wy = py - ay;
    const float l2 = vx * vx + vy * vy;
    float t = l2 > 0.f ? (wx * vx + wy * vy) / l2 : 0.f;
    t = fminf(fmaxf(t, 0.f), 1.f);
    const float dx = wx - t * vx, dy = wy - t * vy;
    return dx * dx + dy * dy;
}

__device__ __forceinline__ void fetch(const uint8_t* img, int H, int W, int y, int x, float* rgb) {
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
        const uint8_t* p = img + ((int64_t)y * W + x) * 3;
        rgb[0] = p[0]; rgb[1] = p[1]; rgb[2] = p[2];
    } else {
        rgb[0] = rgb[1] = rgb[2] = 0.f;
    }
}

// stage 1: compose the defect into a uint8 HWC work image of size h x w
__global__ void compose_kernel(const uint8_t* __restrict__ imgs, const uint8_t* __restrict__ cuts,
                               const ssad_aug_params* __restrict__ params, uint8_t* __restrict__ work, int B, int H, int W,
                               int h, int w) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * h * w) return;
    const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((int64_t)w * h));
    const ssad_aug_params& p = params[b];
    const uint8_t* img = imgs + (int64_t)b * H * W * 3;
    const uint8_t* cut = p.cut_index >= 0 ? cuts + (int64_t)p.cut_index * H * W * 3 : img;
    float rgb[3];
    {   // base pixel: crop window of the (nearest-resampled, zero-filled) affine image
        const float fx = (float)(x + p.crop_left) + 0.5f, fy = (float)(y + p.crop_top) + 0.5f;
        const int sx = (int)floorf(p.aff[0] * fx + p.aff[1] * fy + p.aff[2]);
        const int sy = (int)floorf(p.aff[3] * fx + p.aff[4] * fy + p.aff[5]);
        fetch(img, H, W, sy, sx, rgb);
    }
    if (p.label == 1 && p.patch_w > 0) {
        const int lx = x - p.patch_dst_left, ly = y - p.patch_dst_top;
        if ((unsigned)lx < (unsigned)p.patch_w && (unsigned)ly < (unsigned)p.patch_h &&
            in_polygon(p.poly_xy, p.poly_n, (float)lx + 0.5f, (float)ly + 0.5f)) {
            float s[3];
            if (p.patch_flat) { s[0] = p.patch_rgb[0]; s[1] = p.patch_rgb[1]; s[2] = p.patch_rgb[2]; }
            else fetch(cut, H, W, p.cut_top + p.patch_src_top + ly, p.cut_left + p.patch_src_left + lx, s);
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] = blend_u8(0.f, blend_u8(0.f, s[c], p.patch_bright[0]), p.patch_bright[1]);
        }
    } else if (p.label == 2) {
        for (int k = 0; k < p.scar_n; ++k) {
            const int lx = x - p.scar_dst[2 * k], ly = y - p.scar_dst[2 * k + 1];
            if ((unsigned)lx >= (unsigned)p.scar_rw || (unsigned)ly >= (unsigned)p.scar_rh) continue;
            // inverse rotation about the centres (PIL rotate(angle, expand=True), nearest, transparent outside)
            const float u = (float)lx + 0.5f - 0.5f * (float)p.scar_rw, v = (float)ly + 0.5f - 0.5f * (float)p.scar_rh;
            const int sx = (int)floorf(p.scar_cos * u - p.scar_sin * v + 0.5f * (float)p.scar_w);
            const int sy = (int)floorf(p.scar_sin * u + p.scar_cos * v + 0.5f * (float)p.scar_h);
            if ((unsigned)sx >= (unsigned)p.scar_w || (unsigned)sy >= (unsigned)p.scar_h) continue;
            float s[3];
            if (p.scar_flat) { s[0] = p.scar_rgb[0]; s[1] = p.scar_rgb[1]; s[2] = p.scar_rgb[2]; }
            else fetch(cut, H, W, p.cut_top + p.scar_src_top + sy, p.cut_left + p.scar_src_left + sx, s);
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] = blend_u8(0.f, blend_u8(0.f, s[c], p.scar_bright[0]), p.scar_bright[1]);
        }
    } else if (p.label == 3 && p.line_n > 1) {
        const float r2 = 0.25f * p.line_width * p.line_width;
        const float px = (float)x + 0.5f, py = (float)y + 0.5f;
        for (int k = 0; k + 1 < p.line_n; ++k) {
            if (seg_dist2(px, py, p.line_xy[2 * k] + 0.5f, p.line_xy[2 * k + 1] + 0.5f, p.line_xy[2 * k + 2] + 0.5f,
                          p.line_xy[2 * k + 3] + 0.5f) <= r2) {
                rgb[0] = p.line_rgb[0]; rgb[1] = p.line_rgb[1]; rgb[2] = p.line_rgb[2];
                break;
            }
        }
    }
    uint8_t* o = work + i * 3;
    o[0] = (uint8_t)rgb[0]; o[1] = (uint8_t)rgb[1]; o[2] = (uint8_t)rgb[2];
}

// stage 2: per-sample mean of the L channel (ImageEnhance.Contrast's degenerate image), rounded like PIL
__global__ void gray_mean_kernel(const uint8_t* __restrict__ work, float* __restrict__ mean, int hw) {
    __shared__ double sh[256];
    const uint8_t* p = work + (int64_t)blockIdx.x * hw * 3;
    double s = 0;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) s += (double)gray_u8(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[blockIdx.x] = floorf((float)(sh[0] / hw) + 0.5f);
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
    float r = work[i * 3], g = work[i * 3 + 1], bl = work[i * 3 + 2];
    float mean = gmean[b];
    for (int k = 0; k < 3; ++k) {
        const int op = p.jit_order[k];
        const float f = p.jit_factor[op];
        if (op == 0) {
            r = blend_u8(0.f, r, f); g = blend_u8(0.f, g, f); bl = blend_u8(0.f, bl, f);
            mean = clamp255(mean * f);          // the grey mean seen by a later contrast step scales with it
        } else if (op == 1) {
            r = blend_u8(mean, r, f); g = blend_u8(mean, g, f); bl = blend_u8(mean, bl, f);
        } else {
            const float l = gray_u8(r, g, bl);
            r = blend_u8(l, r, f); g = blend_u8(l, g, f); bl = blend_u8(l, bl, f);
        }
    }
    out[((int64_t)b * 3 + 0) * hw + pix] = (r / 255.f - m0) / s0;
    out[((int64_t)b * 3 + 1) * hw + pix] = (g / 255.f - m1) / s1;
    out[((int64_t)b * 3 + 2) * hw + pix] = (bl / 255.f - m2) / s2;
}

// uint8 HWC -> fp32 CHW in [0,1] (the "original" the Dataset returns as third element)
__global__ void u8_to_f32_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int B, int64_t hw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * hw) return;
    const int b = (int)(i / hw);
    const int64_t pix = i - (int64_t)b * hw;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((int64_t)b * 3 + c) * hw + pix] = (float)img[i * 3 + c] / 255.f;
}

}  // namespace

extern "C" int ssad_aug_params_size(void) { return (int)sizeof(ssad_aug_params); }

extern "C" int ssad_cutpaste_augment(const uint8_t* imgs, const uint8_t* cuts, const ssad_aug_params* params, uint8_t* work,
                                     float* gray_mean, float* out, int B, int H, int W, int h, int w, const float* mean3_host,
                                     const float* std3_host, void* stream) {
    SSAD_CHECK_ARG(imgs && params && work && gray_mean && out && mean3_host && std3_host, "null pointer");
    SSAD_CHECK_ARG(B > 0 && H > 0 && W > 0 && h > 0 && w > 0 && h <= H && w <= W, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)B * h * w;
    hipLaunchKernelGGL(compose_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, imgs, cuts ? cuts : imgs, params,
                       work, B, H, W, h, w);
    hipLaunchKernelGGL(gray_mean_kernel, dim3(B), dim3(256), 0, st, work, gray_mean, h * w);
    hipLaunchKernelGGL(jitter_normalize_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, work, params, gray_mean, out, B,
                       h, w, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    SSAD_CHECK_LAUNCH();
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
