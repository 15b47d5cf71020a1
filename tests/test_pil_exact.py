"""pil_exact.py (the rasterisation rules csrc/augment.hip implements) against the installed Pillow, bit for bit: polygons of
the reference's rect2poly family, poly-lines of width 1 and 3 with float vertices, rotate(expand=True) / RandomAffine-style
nearest transforms, ImageEnhance brightness / contrast / colour.  CPU only."""
import random

import numpy as np
from PIL import Image, ImageDraw, ImageEnhance
from scipy.signal import savgol_filter


def test_polygon_rule_matches_pillow():
    from self_supervised import pil_exact as px
    from self_supervised.dataset_generator import polygon_points
    random.seed(1)
    for it in range(1500):
        w, h = random.randint(2, 60), random.randint(2, 60)
        pts = polygon_points((w, h), sides=8 if it % 5 else 4)
        m = Image.new("L", (w, h), 0)
        ImageDraw.Draw(m).polygon(pts, fill=255)
        got = px.polygon_fill([p[0] for p in pts], [p[1] for p in pts], w, h)
        assert np.array_equal(got, np.array(m) > 0), (w, h, pts)


def test_line_rule_matches_pillow():
    from self_supervised import pil_exact as px
    rng = random.Random(5)
    for width in (1, 3):
        for it in range(500):
            W = H = rng.choice([32, 64, 96])
            if it % 2:
                pts = [(rng.uniform(-3, W + 3), rng.uniform(-3, H + 3)) for _ in range(rng.choice([2, 3, 6, 12]))]
            else:                                  # what datasets.py:357-388 draws: sorted mask pixels, Savitzky-Golay smoothed
                pts = sorted((rng.randint(0, W - 1), rng.randint(0, H - 1)) for _ in range(30))
                pts = savgol_filter(pts, 10, 2, axis=0)
                pts = [tuple(p) for p in np.array_split(pts, 5)[rng.randint(0, 4)]]
            im = Image.new("L", (W, H), 0)
            ImageDraw.Draw(im).line(pts, fill=255, width=width)
            assert np.array_equal(px.draw_line(pts, W, H, width), np.array(im) > 0), (width, W, pts)


def test_affine_and_rotate_match_pillow():
    from self_supervised import pil_exact as px, tv_transforms as tvt
    rng = random.Random(0)
    for it in range(800):
        w, h = rng.randint(2, 40), rng.randint(2, 40)
        arr = np.random.RandomState(it).randint(0, 256, (h, w, 4), dtype=np.uint8)
        arr[..., 3] = 255
        ang = rng.randint(-45, 45)
        ref = np.array(Image.fromarray(arr, "RGBA").rotate(ang, expand=True))
        nw, nh, m = px.rotate_params(w, h, ang)
        got = arr if m is None else px.affine_nearest(arr, (nw, nh), m)
        assert got.shape == ref.shape and np.array_equal(got, ref), (w, h, ang)
    for it in range(300):
        w = h = rng.choice([64, 96, 256, 50])
        arr = np.random.RandomState(it).randint(0, 256, (h, w, 3), dtype=np.uint8)
        m = tvt.inverse_affine_matrix((w * 0.5, h * 0.5), rng.uniform(-3, 3), (0, 0), rng.uniform(1.05, 1.1))
        ref = np.array(Image.fromarray(arr, "RGB").transform((w, h), Image.AFFINE, m, Image.NEAREST))
        assert np.array_equal(px.affine_nearest(arr, (w, h), m), ref)


def test_enhance_matches_pillow():
    from self_supervised import pil_exact as px
    rng = random.Random(3)
    enh = (ImageEnhance.Brightness, ImageEnhance.Contrast, ImageEnhance.Color)
    for it in range(400):
        h, w = rng.randint(2, 40), rng.randint(2, 40)
        arr = np.random.RandomState(it).randint(0, 256, (h, w, 3), dtype=np.uint8)
        if it % 3 == 0:
            arr = (arr // 4 + 180).astype(np.uint8)              # bright: the extrapolating factors clamp
        im = Image.fromarray(arr, "RGB")
        for op in range(3):
            f = rng.uniform(0.75, 1.15) if it % 5 == 0 else float(np.float32(rng.uniform(0.7, 1.2)))
            assert np.array_equal(px.ENHANCERS[op](arr, f), np.array(enh[op](im).enhance(f))), (op, f)


def test_bicubic_resize_tables_match_pillow():
    """pil_exact.resample_coeffs / resize_bicubic (the tables csrc/resize.hip consumes and the numpy statement of its two integer
    passes) against Image.resize of the installed Pillow: 'L' and 'RGB', down- and up-scaling, integer and fractional ratios,
    one extent unchanged."""
    from PIL import Image
    from self_supervised import pil_exact as px
    rng = np.random.RandomState(0)
    cases = [((700, 700), (256, 256)), ((1024, 1024), (256, 256)), ((900, 900), (256, 256)), ((256, 256), (64, 64)),
             ((100, 130), (256, 256)), ((257, 255), (256, 256)), ((840, 1000), (320, 200)), ((256, 300), (256, 256)),
             ((300, 256), (256, 256)), ((33, 47), (64, 64)), ((96, 96), (256, 256))]
    for (h, w), (oh, ow) in cases:
        for c in (1, 3):
            noise = rng.randint(0, 256, (h, w, c) if c == 3 else (h, w)).astype(np.uint8)
            yy, xx = np.mgrid[0:h, 0:w]
            smooth = (127 + 120 * np.sin(xx / 9.0) * np.cos(yy / 13.0)).astype(np.uint8)
            smooth = smooth if c == 1 else np.stack([smooth, 255 - smooth, smooth // 2], -1)
            for img in (noise, smooth):
                want = np.asarray(Image.fromarray(img).resize((ow, oh)))
                assert np.array_equal(px.resize_bicubic(img, (ow, oh)), want), ((h, w), (oh, ow), c)
    ks, bounds, kk = px.resample_coeffs(1024, 256)
    assert ks == 17 and bounds.shape == (256, 2) and kk.shape == (256, 17) and int(kk.sum(1).min()) > (1 << 22) - 16
