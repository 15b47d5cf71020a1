"""Cut-paste primitives on PIL images (host side of the synthetic-defect generator).

Mirrors src/self_supervised/dataset_generator.py:15-275 of the reference: same function names, arguments and --
because results are compared against vectors produced by the reference under a fixed ``random.seed`` -- the same
order of draws from Python's ``random``.  The batched GPU version of the pixel work (polygon rasterise, masked
paste, rotated scars, poly-lines, jitter, normalise) is csrc/augment.hip, driven by augment.py.
"""
import random
from typing import Tuple

import numpy as np
from PIL import Image, ImageDraw
from scipy import ndimage


class Container:
    """Centred square region defects are clamped into (dataset_generator.py:15-24)."""

    def __init__(self, imsize: tuple, scaling_factor: float) -> None:
        half = int(imsize[0] / 2)
        reach = half / scaling_factor
        self.center = half
        self.dim = int(imsize[0] / scaling_factor)
        self.left = self.top = int(half - reach)
        self.right = self.bottom = int(half + reach)
        self.width = self.right - self.left
        self.height = self.bottom - self.top


# ---- object mask: Canny -> dilate -> close -> fill -> erode -> largest component (dataset_generator.py:27-39) ----
def _canny(gray, sigma, low, high):
    """skimage.feature.canny(gray_uint8, sigma, low_threshold=low, high_threshold=high) -- THIRD-PARTY RESTATEMENT, unpinned
    (scikit-image is not installed): its documented pipeline, step for step -- the image scaled to [0, 1] (thresholds divided
    by 255 accordingly), Gaussian smoothing with zero ('constant') borders renormalised by the smoothed all-ones mask, Sobel
    derivatives, non-maximum suppression by bilinear interpolation of the magnitude along the gradient direction on the
    border-eroded mask, double threshold and hysteresis over 8-connected components."""
    img = gray.astype(np.float64) / 255.0
    low, high = low / 255.0, high / 255.0
    bleed = ndimage.gaussian_filter(np.ones_like(img), sigma, mode="constant", cval=0.0) + np.finfo(np.float64).eps
    sm = ndimage.gaussian_filter(img, sigma, mode="constant", cval=0.0) / bleed
    js = ndimage.sobel(sm, axis=1)
    is_ = ndimage.sobel(sm, axis=0)
    mag = np.hypot(is_, js)
    h, w = mag.shape
    out = np.zeros_like(mag)
    if h < 3 or w < 3:
        return np.zeros(mag.shape, bool)
    c = (slice(1, -1), slice(1, -1))

    def sh(dx, dy):                                  # magnitude at (row + dx, col + dy) for the interior pixels
        return mag[1 + dx:h - 1 + dx, 1 + dy:w - 1 + dy]
    i, j, m = is_[c], js[c], mag[c]
    ai, aj = np.abs(i), np.abs(j)
    ok = m >= low                                    # the eroded mask only removes the one-pixel border, i.e. the interior
    up, down, left, right = i >= 0, i <= 0, j <= 0, j >= 0
    c1 = (up & right) | (down & left)
    c2 = ~c1 & ((down & right) | (up & left))
    with np.errstate(divide="ignore", invalid="ignore"):
        w_ji, w_ij = np.where(ai > 0, aj / ai, 0.0), np.where(aj > 0, ai / aj, 0.0)
    keep = np.zeros(m.shape, bool)

    def test(sel, wgt, n11, n12, n21, n22):
        plus = (n12 * wgt + n11 * (1.0 - wgt)) <= m
        minus = (n22 * wgt + n21 * (1.0 - wgt)) <= m
        keep[sel & plus & minus] = True
    a = c1 & (ai > aj)
    test(a, w_ji, sh(1, 0), sh(1, 1), sh(-1, 0), sh(-1, -1))
    test(c1 & ~a, w_ij, sh(0, 1), sh(1, 1), sh(0, -1), sh(-1, -1))
    b = c2 & (ai < aj)
    test(b, w_ij, sh(0, 1), sh(-1, 1), sh(0, -1), sh(1, -1))
    test(c2 & ~b, w_ji, sh(-1, 0), sh(-1, 1), sh(1, 0), sh(1, -1))
    out[c] = np.where(keep & ok, m, 0.0)
    low_mask = out > 0
    lab, n = ndimage.label(low_mask, structure=np.ones((3, 3), int))
    if n == 0:
        return low_mask
    high_mask = low_mask & (out >= high)
    sums = ndimage.sum_labels(high_mask, lab, np.arange(n + 1))
    good = sums > 0
    good[0] = False
    return good[lab]


def obj_mask(image: Image.Image) -> Image.Image:
    gray = np.array(image.convert('L'))
    edges = _canny(gray, sigma=1.5, low=5, high=15)
    sq3, sq4 = np.ones((3, 3), int), np.ones((4, 4), int)
    m = ndimage.binary_dilation(edges, sq3)
    m = ndimage.binary_closing(m, sq3)
    m = ndimage.binary_fill_holes(m, sq3)
    m = ndimage.binary_erosion(m, sq4)
    lab, _ = ndimage.label(m, structure=np.ones((3, 3), int))      # skimage.morphology.label: full (8-) connectivity by default
    # no component at all: bincount = [0], argmax 0, `labels == 0` is everywhere true -- the reference's white mask
    sizes = np.bincount(lab.ravel(), weights=m.ravel().astype(np.float64))
    return Image.fromarray(lab == int(np.argmax(sizes))).convert('RGB')


# ---- polygon mask (dataset_generator.py:42-101) ----
def _side_points(side: int, w: int, h: int):
    """One or two points on a side of the w x h rectangle, drawn in the reference's order."""
    two = random.randint(1, 2) == 2
    hw, hh = int(w / 2), int(h / 2)
    if side == 0:      # left, walking upwards
        return [(0, random.randint(hh + 1, h)), (0, random.randint(1, hh))] if two else [(0, random.randint(1, h))]
    if side == 1:      # top, walking right
        return [(random.randint(1, hw), 0), (random.randint(hw + 1, w), 0)] if two else [(random.randint(1, w), 0)]
    if side == 2:      # right, walking down
        return [(w, random.randint(1, hh)), (w, random.randint(hh + 1, h))] if two else [(w, random.randint(1, h))]
    return [(random.randint(hw + 1, w), h), (random.randint(1, hw), h)] if two else [(random.randint(1, w), h)]


def polygon_points(size: Tuple[int, int], sides=4):
    """Vertex list of the irregular polygon inscribed in a ``size`` rectangle (4 or up to 8 vertices)."""
    w, h = size
    if sides == 4:
        return [(0, random.randint(1, h)), (random.randint(1, w), 0), (w, random.randint(1, h)), (random.randint(1, w), h)]
    pts = []
    for side in range(4):
        pts += _side_points(side, w, h)
    return pts


def rect2poly(patch: Image.Image, regular: bool = False, sides: list = 4):
    """RGBA mask (white polygon on transparent) of the patch's size."""
    w, h = patch.size
    mask = Image.new('RGBA', patch.size, color=(0, 0, 0, 0))
    draw = ImageDraw.Draw(mask)
    if regular:
        draw.regular_polygon(bounding_circle=((int(w / 2), int(h / 2)), int(min(w, h) / 2)),
                             n_sides=random.choice(sides), fill='white')
    else:
        draw.polygon(polygon_points((w, h), sides), fill='white')
    return mask


# ---- placement (dataset_generator.py:104-144) ----
def check_valid_coordinates_by_container(imsize: tuple, patchsize: tuple, current_coords: tuple = None,
                                         container_scaling_factor: int = 1):
    """Top-left paste corner for a patch centred at ``current_coords`` (random centre when None), pulled back
    inside the container on each side in the order right, bottom, left, top."""
    pw, ph = patchsize
    box = Container(imsize, scaling_factor=container_scaling_factor)
    if current_coords is None:
        cx, cy = random.randint(box.left, box.right), random.randint(box.top, box.bottom)
    else:
        cx, cy = current_coords[0], current_coords[1]
    left, top = cx - int(pw / 2), cy - int(ph / 2)
    right, bottom = cx + int(pw / 2), cy + int(ph / 2)
    if right > box.right:
        left = box.right - pw
    if bottom > box.bottom:
        top = box.bottom - ph
    if left < box.left:
        left = box.left
    if top < box.top:
        top = box.top
    return (left, top)


def check_color_similarity(patch: Image.Image, defect: Image.Image) -> float:
    """Cosine similarity of the two mean RGB colours (dataset_generator.py:147-159)."""
    a = np.array(patch).mean(axis=(0, 1))[:3] / 255.0
    b = np.array(defect).mean(axis=(0, 1))[:3] / 255.0
    return float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b)))


# ---- patch sampling (dataset_generator.py:164-210) ----
def sample_patch_box(imsize: Tuple[int, int], area_ratio, aspect_ratio):
    """(left, top, w, h) of the random rectangle: area ~ U(area_ratio) * image area, aspect from one of two ranges."""
    iw, ih = imsize
    area = random.uniform(area_ratio[0], area_ratio[1]) * (iw * ih)
    aspect = random.choice([random.uniform(*aspect_ratio[0]), random.uniform(*aspect_ratio[1])])
    pw, ph = max(int(np.sqrt(area * aspect)), 2), max(int(np.sqrt(area / aspect)), 2)
    left, top = random.randint(0, max(iw - pw, 1)), random.randint(0, max(ih - ph, 1))
    return left, top, pw, ph


def generate_patch(image: Image.Image, area_ratio: tuple = (0.02, 0.15), aspect_ratio: tuple = ((0.3, 1), (1, 3.3)),
                   augs=None, colorized: bool = False, color_type: str = 'random') -> Image.Image:
    left, top, pw, ph = sample_patch_box(image.size, area_ratio, aspect_ratio)
    box = (left, top, left + pw, top + ph)
    if not colorized:
        return image.crop(box)
    if color_type == 'random':
        rgb = (random.randint(0, 255), random.randint(0, 255), random.randint(0, 255))
    elif color_type == 'sample':
        rgb = random.choice(['black', 'white', 'silver', 'gray'])
    else:                                                  # 'average'
        mean = np.array(image.crop(box)).mean(axis=(0, 1))
        rgb = (int(mean[0]), int(mean[1]), int(mean[2]))
    return Image.new('RGB', (pw, ph), color=rgb)


def get_random_coordinate(xy_coords) -> Tuple[int, int]:
    if len(xy_coords) == 0:
        return None
    if len(xy_coords) < 2:
        return xy_coords[0]
    return xy_coords[random.randint(0, len(xy_coords) - 1)]


def paste_patch(image: Image.Image, patch: Image.Image, coords: tuple, mask: Image.Image = None):
    out = image.copy()
    out.paste(patch, (coords[0], coords[1]), mask=mask)
    return out


# ---- 'cable' pre-segmentation: SLIC super-pixels + per-segment mean colour (datasets.py:201-206) ----
# THIRD-PARTY RESTATEMENT, unpinned: the reference calls skimage.segmentation.slic(image, n_segments=5, sigma=2,
# convert2lab=True) and skimage.color.label2rgb(segments, image, kind='avg'); scikit-image is not installed, so the
# published algorithm (Achanta et al., "SLIC Superpixels", as scikit-image documents its implementation: Gaussian
# pre-smoothing, CIELAB, k-means in (L, a, b, y, x) restricted to 2S x 2S windows with compactness 10, 10 iterations,
# connectivity enforcement with min / max size factors 0.5 / 3, labels from 1) is written out here.  It yields the same kind
# of result -- a handful of compact colour regions -- not scikit-image's exact label image.
def _rgb2lab(rgb):
    x = rgb.astype(np.float64) / 255.0
    lin = np.where(x > 0.04045, ((x + 0.055) / 1.055) ** 2.4, x / 12.92)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = lin @ m.T / np.array([0.95047, 1.0, 1.08883])
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    return np.stack([116.0 * f[..., 1] - 16.0, 500.0 * (f[..., 0] - f[..., 1]), 200.0 * (f[..., 1] - f[..., 2])], axis=-1)


def slic_superpixels(image_array, n_segments=5, sigma=2, compactness=10.0, max_num_iter=10):
    """Label image (int, starting at 1) of ~n_segments SLIC super-pixels of an H x W x 3 uint8 array."""
    h, w = image_array.shape[:2]
    lab = _rgb2lab(image_array)
    lab = np.stack([ndimage.gaussian_filter(lab[..., c], sigma, mode="nearest") for c in range(3)], axis=-1)
    # regular grid of initial centres: ~n_segments cells of equal area
    step = max(int(round(np.sqrt(h * w / float(n_segments)))), 1)
    gy = np.arange(step // 2, h, step)
    gx = np.arange(step // 2, w, step)
    if len(gy) == 0:
        gy = np.array([h // 2])
    if len(gx) == 0:
        gx = np.array([w // 2])
    cy, cx = [a.ravel().astype(np.float64) for a in np.meshgrid(gy, gx, indexing="ij")]
    centres = np.concatenate([lab[cy.astype(int), cx.astype(int)], cy[:, None], cx[:, None]], axis=1)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    ratio = (compactness / float(step)) ** 2                        # spatial weight of D^2 = d_lab^2 + ratio * d_xy^2
    labels = np.zeros((h, w), np.int64)
    for _ in range(max_num_iter):
        best = np.full((h, w), np.inf)
        for k, (l, a, b, y0, x0) in enumerate(centres):
            ya, yb = max(int(y0 - 2 * step), 0), min(int(y0 + 2 * step) + 1, h)
            xa, xb = max(int(x0 - 2 * step), 0), min(int(x0 + 2 * step) + 1, w)
            win = lab[ya:yb, xa:xb]
            d = ((win - np.array([l, a, b])) ** 2).sum(-1) + ratio * ((yy[ya:yb, xa:xb] - y0) ** 2 + (xx[ya:yb, xa:xb] - x0) ** 2)
            sub = best[ya:yb, xa:xb]
            better = d < sub
            sub[better] = d[better]
            labels[ya:yb, xa:xb][better] = k
        moved = False
        for k in range(len(centres)):
            msk = labels == k
            if msk.any():
                new = np.concatenate([lab[msk].mean(0), [yy[msk].mean(), xx[msk].mean()]])
                moved = moved or not np.allclose(new, centres[k])
                centres[k] = new
        if not moved:
            break
    # connectivity: every 4-connected component smaller than half the nominal segment joins its most frequent neighbour
    out = np.zeros((h, w), np.int64)
    nxt = 0
    for k in range(len(centres)):
        c, m = ndimage.label(labels == k)
        out[c > 0] = c[c > 0] + nxt
        nxt += m
    min_size = int(0.5 * h * w / max(len(centres), 1))
    sizes = np.bincount(out.ravel(), minlength=nxt + 1)
    order = [lab_id for lab_id in range(1, nxt + 1) if 0 < sizes[lab_id] < min_size]
    for lab_id in sorted(order, key=lambda i: sizes[i]):
        msk = out == lab_id
        ring = ndimage.binary_dilation(msk) & ~msk
        neigh = out[ring]
        neigh = neigh[neigh != lab_id]
        if len(neigh):
            target = np.bincount(neigh).argmax()
            out[msk] = target
            sizes[target] += sizes[lab_id]
            sizes[lab_id] = 0
    _, dense = np.unique(out, return_inverse=True)
    return dense.reshape(h, w) + 1


def label_mean_rgb(segments, image_array):
    """skimage.color.label2rgb(segments, image, kind='avg'): every pixel takes the mean colour of its segment (uint8)."""
    out = np.zeros_like(image_array)
    for s in np.unique(segments):
        msk = segments == s
        out[msk] = image_array[msk].mean(axis=0).astype(image_array.dtype)
    return out
