"""Exact restatements of the Pillow rasterisation rules the reference's augmentation relies on
(src/self_supervised/datasets.py:209-394, dataset_generator.py:42-101 call ImageDraw.polygon / ImageDraw.line /
Image.rotate(expand=True) / Image.transform(AFFINE) / Image.paste / ImageEnhance on uint8 images).

Pillow is a third-party dependency of the reference; what is restated here is its PUBLISHED algorithm (libImaging
Draw.c `polygon_generic` / `ImagingDrawWideLine` / `line8`, Geometry.c `affine_fixed` / `ImagingScaleAffine`, Blend.c,
Convert.c `rgb2l`, Image.rotate in Image.py), pinned bit-for-bit against the installed Pillow by
tests/test_pil_exact.py on thousands of random cases.  Two users:
  * augment.sample_defect -- the numbers the GPU kernel needs (16.16 fixed-point affine coefficients, rotated sizes,
    truncated poly-line points, wide-line quads) are computed here exactly as Pillow computes them;
  * the numpy rasterisers below are the CPU statement of what csrc/augment.hip does per pixel (same rules, same float32
    operation order), used by the tests next to Pillow itself.
"""
import math

import numpy as np

f32 = np.float32


# ---------------------------------------------------------------------------------------------
# affine / rotate, NEAREST (Geometry.c affine_fixed: 16.16 fixed point)
# ---------------------------------------------------------------------------------------------
def fix16(v):
    """FIX(v) = FLOOR(v * 65536.0 + 0.5)"""
    return int(math.floor(v * 65536.0 + 0.5))


def affine_fix_coeffs(a):
    """The six integers affine_fixed iterates with: source x = (a2 + y*a1 + x*a0) >> 16, y = (a5 + y*a4 + x*a3) >> 16
    (the half-pixel centre offset is folded into a2 / a5).  `a` = the six doubles of Image.transform(AFFINE)."""
    return (fix16(a[0]), fix16(a[1]), fix16(a[2] + a[0] * 0.5 + a[1] * 0.5),
            fix16(a[3]), fix16(a[4]), fix16(a[5] + a[3] * 0.5 + a[4] * 0.5))


def affine_is_scale_only(a):
    """a[1] == a[3] == 0 takes Pillow's ImagingScaleAffine path (floating-point row / column tables) instead."""
    return a[1] == 0 and a[3] == 0


def affine_nearest(src, out_size, a):
    """Image.transform(out_size, AFFINE, a, NEAREST) on an H x W x C uint8 array (zero fill)."""
    ow, oh = out_size
    h, w = src.shape[:2]
    out = np.zeros((oh, ow) + src.shape[2:], src.dtype)
    if affine_is_scale_only(a):
        xo, yo = a[2] + a[0] * 0.5, a[5] + a[4] * 0.5
        xin = np.empty(ow, np.int64)
        for x in range(ow):
            xin[x] = -1 if xo < 0.0 else int(xo)
            xo += a[0]
        okx = (xin >= 0) & (xin < w)
        for y in range(oh):
            yi = -1 if yo < 0.0 else int(yo)
            if 0 <= yi < h:
                out[y, okx] = src[yi, xin[okx]]
            yo += a[4]
        return out
    a0, a1, a2, a3, a4, a5 = affine_fix_coeffs(a)
    ys, xs = np.mgrid[0:oh, 0:ow].astype(np.int64)
    xx = (a2 + ys * a1 + xs * a0) >> 16
    yy = (a5 + ys * a4 + xs * a3) >> 16
    ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
    out[ok] = src[yy[ok], xx[ok]]
    return out


def rotate_params(w, h, angle):
    """Image.rotate(angle, expand=True) -> (new_w, new_h, matrix) with matrix None on the copy fast path (angle % 360
    == 0); multiples of 90 other than 0 are transposes in Pillow and are not needed here (|angle| <= 45)."""
    angle = angle % 360.0
    if angle == 0:
        return w, h, None
    assert angle not in (90, 180, 270), "transpose fast paths are not restated"
    cx, cy = w / 2, h / 2
    ang = -math.radians(angle)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]

    def tr(x, y):
        return m[0] * x + m[1] * y + m[2], m[3] * x + m[4] * y + m[5]
    m[2], m[5] = tr(-cx, -cy)
    m[2] += cx
    m[5] += cy
    xs, ys = zip(*(tr(x, y) for x, y in ((0, 0), (w, 0), (w, h), (0, h))))
    nw = math.ceil(max(xs)) - math.floor(min(xs))
    nh = math.ceil(max(ys)) - math.floor(min(ys))
    m[2], m[5] = tr(-(nw - w) / 2.0, -(nh - h) / 2.0)
    return nw, nh, m


# ---------------------------------------------------------------------------------------------
# polygon / line (Draw.c)
# ---------------------------------------------------------------------------------------------
def round_up(f):
    """ROUND_UP: (int)(f >= 0 ? floor(f + 0.5F) : -floor(fabs(f) + 0.5F))"""
    f = float(f)
    return int(math.floor(f + 0.5)) if f >= 0.0 else -int(math.floor(abs(f) + 0.5))


def round_down(f):
    """ROUND_DOWN: (int)(f >= 0 ? ceil(f - 0.5F) : -ceil(fabs(f) - 0.5F))"""
    f = float(f)
    return int(math.ceil(f - 0.5)) if f >= 0.0 else -int(math.ceil(abs(f) - 0.5))


def polygon_row_spans(vx, vy, W, H, y):
    """Inclusive [x0, x1] pixel spans that ImageDraw.polygon(fill) paints on row y of a W x H image for integer vertices
    (polygon_generic: horizontal edges are drawn as they are; the other edges are intersected with the scan line in float32
    as (y - y0) * dx + x0, an intersection at an edge's lower end is doubled unless it is the polygon's last row, the sorted
    intersections are painted pairwise from ROUND_UP(left) to ROUND_DOWN(right) without going back over painted pixels).
    The corner-joining branch of newer Pillow versions only fires for shapes this module never draws (checked by the tests)."""
    n = len(vx)
    spans = []
    pymin, pymax = H - 1, 0
    edges = []
    for i in range(n):
        x0, y0, x1, y1 = int(vx[i]), int(vy[i]), int(vx[(i + 1) % n]), int(vy[(i + 1) % n])
        ymin, ymax = min(y0, y1), max(y0, y1)
        pymin, pymax = min(pymin, ymin), max(pymax, ymax)
        if y0 == y1:
            if y0 == y:
                spans.append((min(x0, x1), max(x0, x1)))
            continue
        edges.append((x0, y0, ymin, ymax, f32(x1 - x0) / f32(y1 - y0)))
    pymin, pymax = max(pymin, 0), min(pymax, H)
    if pymin <= y <= pymax:
        xx = []
        for x0, y0, ymin, ymax, dx in edges:
            if ymin <= y <= ymax:
                xx.append(f32(f32(y - y0) * dx) + f32(x0))
                if y == ymax and y < pymax:
                    xx.append(xx[-1])
        xx.sort()
        x_pos = int(xx[0]) if xx else 0
        for i in range(1, len(xx), 2):
            x_end = round_down(xx[i])
            if x_end < x_pos:
                continue
            x_start = round_up(xx[i - 1])
            if x_pos > x_start:
                x_start = x_pos
                if x_end < x_start:
                    continue
            spans.append((x_start, x_end))
            x_pos = x_end + 1
    return [(max(a, 0), min(b, W - 1)) for a, b in spans if max(a, 0) <= min(b, W - 1)] if 0 <= y < H else []


def polygon_fill(vx, vy, W, H):
    """ImageDraw.polygon(list(zip(vx, vy)), fill=...) as an H x W boolean mask."""
    out = np.zeros((H, W), bool)
    for y in range(H):
        for a, b in polygon_row_spans(vx, vy, W, H, y):
            out[y, a:b + 1] = True
    return out


def line_points_int(points):
    """ImageDraw.line hands (int) casts of the coordinates to the rasteriser: truncation toward zero."""
    return [(int(p[0]), int(p[1])) for p in points]


def wide_line_quad(x0, y0, x1, y1, width):
    """ImagingDrawWideLine: the four integer vertices of the quadrilateral a `width` > 1 segment is painted as, or None
    for a degenerate (single point) segment."""
    dx, dy = x1 - x0, y1 - y0
    if dx == 0 and dy == 0:
        return None
    big = math.hypot(dx, dy)
    small = (width - 1) / 2.0
    rmax, rmin = round_up(small) / big, round_down(small) / big
    dxmin, dxmax = round_down(rmin * dy), round_down(rmax * dy)
    dymin, dymax = round_up(rmin * dx), round_up(rmax * dx)
    return [(x0 - dxmin, y0 + dymax), (x1 - dxmin, y1 + dymax), (x1 + dxmax, y1 - dymin), (x0 + dxmax, y0 - dymin)]


def bresenham(out, x0, y0, x1, y1):
    """Draw.c line8/line32: every pixel of the segment EXCEPT its end point."""
    H, W = out.shape

    def pt(x, y):
        if 0 <= x < W and 0 <= y < H:
            out[y, x] = True
    dx, xs = x1 - x0, 1
    if dx < 0:
        dx, xs = -dx, -1
    dy, ys = y1 - y0, 1
    if dy < 0:
        dy, ys = -dy, -1
    if dx == 0:
        for _ in range(dy):
            pt(x0, y0); y0 += ys
    elif dy == 0:
        for _ in range(dx):
            pt(x0, y0); x0 += xs
    elif dx > dy:
        n = dx; dy += dy; e = dy - dx; dx += dx
        for _ in range(n):
            pt(x0, y0)
            if e >= 0:
                y0 += ys; e -= dx
            e += dy; x0 += xs
    else:
        n = dy; dx += dx; e = dx - dy; dy += dy
        for _ in range(n):
            pt(x0, y0)
            if e >= 0:
                x0 += xs; e -= dy
            e += dx; y0 += ys


def draw_line(points, W, H, width):
    """ImageDraw.line(points, fill=..., width=width) as an H x W boolean mask."""
    out = np.zeros((H, W), bool)
    ip = line_points_int(points)
    if width <= 1:
        for (x0, y0), (x1, y1) in zip(ip[:-1], ip[1:]):
            bresenham(out, x0, y0, x1, y1)
        if len(ip) > 1 and 0 <= ip[-1][0] < W and 0 <= ip[-1][1] < H:
            out[ip[-1][1], ip[-1][0]] = True
        return out
    for (x0, y0), (x1, y1) in zip(ip[:-1], ip[1:]):
        q = wide_line_quad(x0, y0, x1, y1, width)
        if q is None:
            if 0 <= x0 < W and 0 <= y0 < H:
                out[y0, x0] = True
            continue
        out |= polygon_fill([v[0] for v in q], [v[1] for v in q], W, H)
    return out


# ---------------------------------------------------------------------------------------------
# ImageEnhance (Blend.c, Convert.c)
# ---------------------------------------------------------------------------------------------
def to_L(rgb):
    """convert('L'): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16"""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(degenerate, img, alpha):
    """Image.blend(degenerate, img, alpha): float32 `in1 + alpha * (in2 - in1)` truncated to uint8; clamped to [0, 255]
    when alpha extrapolates (outside [0, 1])."""
    a = f32(alpha)
    d, v = degenerate.astype(np.int32), img.astype(np.int32)
    t = d.astype(f32) + a * (v - d).astype(f32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t)).astype(np.uint8)


def enhance_brightness(img, f):
    return blend(np.zeros_like(img), img, f)


def gray_mean(img):
    """int(ImageStat.Stat(img.convert('L')).mean[0] + 0.5)"""
    L = to_L(img)
    return int(float(L.astype(np.int64).sum()) / L.size + 0.5)


def enhance_contrast(img, f):
    return blend(np.full_like(img, gray_mean(img)), img, f)


def enhance_color(img, f):
    return blend(np.repeat(to_L(img)[..., None], 3, axis=2), img, f)


ENHANCERS = (enhance_brightness, enhance_contrast, enhance_color)       # indexed like ColorJitter's ops


# ---------------------------------------------------------------------------------------------
# Image.resize(size) with Pillow's default filter for 'L' / 'RGB' images, BICUBIC (libImaging Resample.c, 8 bits per channel:
# precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) -- what the reference applies to
# every image it opens (src/self_supervised/datasets.py:68, :211-213, functional.py:20-25).  csrc/resize.hip runs the two integer
# passes on the device from the tables computed here.
# ---------------------------------------------------------------------------------------------
RESAMPLE_PRECISION_BITS = 32 - 8 - 2


def _bicubic_filter(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size):
    """One axis of Image.resize (box = the whole image): -> (ksize, bounds int32 [out][2] = (first input index, taps), coefficients
    int32 [out][ksize] in 22-bit fixed point), exactly as precompute_coeffs + normalize_coeffs_8bpc compute them (C doubles)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << RESAMPLE_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << RESAMPLE_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _resample_axis(img, bounds, kk, axis):
    """One pass over an H x W x C uint8 array (numpy statement of the kernel: int32 sums, rounding constant, arithmetic shift, clip)."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.int64)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (RESAMPLE_PRECISION_BITS - 1), np.int64)
        for t in range(n):
            acc += src[x0 + t] * int(kk[xx, t])
        out[xx] = np.clip(acc >> RESAMPLE_PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize_bicubic(img, size):
    """Image.fromarray(img).resize(size) for an H x W (mode 'L') or H x W x 3 (mode 'RGB') uint8 array; size = (width, height).
    Horizontal pass first, then vertical, each skipped when that extent does not change (ImagingResample)."""
    a = img if img.ndim == 3 else img[:, :, None]
    w_out, h_out = size
    if a.shape[1] != w_out:
        _, b, k = resample_coeffs(a.shape[1], w_out)
        a = _resample_axis(a, b, k, 1)
    if a.shape[0] != h_out:
        _, b, k = resample_coeffs(a.shape[0], h_out)
        a = _resample_axis(a, b, k, 0)
    return a if img.ndim == 3 else a[:, :, 0]
