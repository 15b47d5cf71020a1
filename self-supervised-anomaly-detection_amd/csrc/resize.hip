// Image.resize(size) of Pillow for 8-bit 'L' / 'RGB' images with its default filter (BICUBIC), on the device.
//
// Replaces the `Image.open(f).resize(imsize)` the reference applies to every image it reads
// (src/self_supervised/datasets.py:68 MVTecDataset.__getitem__, :211-213 PretextTaskDataset.__getitem__, :189-200 the cut
// sources; functional.py:20-25 the ground-truth masks) for batches that are already in HBM as uint8: the loaders upload the decoded
// files at their native size and resize there.  libImaging/Resample.c, 8 bits per channel: two separable passes (horizontal, then
// vertical on the clipped uint8 result of the first), every output a sum of at most `ksize` inputs times 22-bit fixed-point
// coefficients, + 2^21, arithmetic shift by 22, clipped to [0, 255].  The coefficient and bound tables come from the host
// (self_supervised/pil_exact.resample_coeffs: Pillow's double-precision arithmetic, pinned against the installed Pillow), so the
// kernels are pure integer work and bit-exact.  HBM-bound: one read of the source, one write of the result (+ the Hin x Wout
// intermediate).
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;                    // arithmetic shift (Pillow indexes a table that starts at -640)
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one workgroup = one input row of one image: the row goes through LDS, every thread forms outputs (xx, c) from it
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int Hin, int Win,
                                                       int C, int Wout, const int32_t* __restrict__ bounds,
                                                       const int32_t* __restrict__ coef, int ksize) {
    extern __shared__ uint8_t row[];
    const int y = blockIdx.x, b = blockIdx.y;
    const uint8_t* src = in + ((int64_t)b * Hin + y) * Win * C;
    const int nbytes = Win * C;
    for (int i = threadIdx.x; i < nbytes; i += 256) row[i] = src[i];
    __syncthreads();
    uint8_t* dst = out + ((int64_t)b * Hin + y) * Wout * C;
    for (int o = threadIdx.x; o < Wout * C; o += 256) {
        const int xx = o / C, c = o - xx * C;
        const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
        const int32_t* k = coef + (int64_t)xx * ksize;
        int acc = 1 << (PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) acc += (int)row[(x0 + t) * C + c] * k[t];
        dst[o] = clip8(acc);
    }
}

// one thread = one output byte (yy, xx, c): its taps are the same byte of consecutive source rows (coalesced across the wave)
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int Hin, int Hout,
                                                       int rowbytes, const int32_t* __restrict__ bounds,
                                                       const int32_t* __restrict__ coef, int ksize) {
    const int yy = blockIdx.y, b = blockIdx.z;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= rowbytes) return;
    const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int32_t* k = coef + (int64_t)yy * ksize;
    const uint8_t* src = in + ((int64_t)b * Hin + y0) * rowbytes + o;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int t = 0; t < n; ++t) acc += (int)src[(int64_t)t * rowbytes] * k[t];
    out[((int64_t)b * Hout + yy) * rowbytes + o] = clip8(acc);
}

}  // namespace

extern "C" int ssad_resize_bicubic_u8(const uint8_t* in, uint8_t* tmp, uint8_t* out, int B, int Hin, int Win, int C, int Hout,
                                      int Wout, const int32_t* bounds_x, const int32_t* coef_x, int ksize_x,
                                      const int32_t* bounds_y, const int32_t* coef_y, int ksize_y, void* stream) {
    SSAD_CHECK_ARG(in && out && B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && C >= 1 && C <= 4, "bad argument");
    const bool horiz = Win != Wout, vert = Hin != Hout;
    SSAD_CHECK_ARG(horiz || vert, "nothing to resize (Image.resize returns a copy): do not call");
    SSAD_CHECK_ARG(!horiz || (bounds_x && coef_x && ksize_x > 0), "horizontal tables missing");
    SSAD_CHECK_ARG(!vert || (bounds_y && coef_y && ksize_y > 0), "vertical tables missing");
    SSAD_CHECK_ARG(!(horiz && vert) || tmp, "both passes need the Hin x Wout intermediate");
    SSAD_CHECK_ARG((int64_t)Win * C <= 60 * 1024, "row too long for the LDS row buffer");
    SSAD_CHECK_ARG(B <= 65535 && Hout <= 65535, "grid limit");
    hipStream_t st = (hipStream_t)stream;
    if (horiz) {
        uint8_t* dst = vert ? tmp : out;
        hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)Hin, (unsigned)B), dim3(256), (size_t)Win * C, st, in, dst, Hin, Win, C, Wout,
                           bounds_x, coef_x, ksize_x);
        SSAD_CHECK_LAUNCH();
    }
    if (vert) {
        const uint8_t* src = horiz ? tmp : in;
        const int rowbytes = Wout * C;
        hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((rowbytes + 255) / 256), (unsigned)Hout, (unsigned)B), dim3(256), 0, st, src,
                           out, Hin, Hout, rowbytes, bounds_y, coef_y, ksize_y);
        SSAD_CHECK_LAUNCH();
    }
    return 0;
}
