#!/opt/conda/bin/python3.9
"""Golden vectors for the scikit-image calls of the reference (dataset_generator.py:27-39 obj_mask -> feature.canny /
morphology.square / label; datasets.py:203-204 slic + label2rgb for 'cable').

scikit-image is not installed for the interpreter the framework runs on, but the build container carries a conda Python 3.9
with scikit-image 0.18.3 / scipy 1.7.1 / numpy 1.26.4 (/opt/conda).  Run THERE:

    /opt/conda/bin/python3.9 tests/golden/make_skimage_fixtures.py

It imports the REFERENCE's own `dataset_generator` (torchvision's ColorJitter, the only missing import, is stubbed -- obj_mask
never touches it) and records, for a set of synthetic images that travel inside the fixture: skimage's canny edge map of the
gray image, the reference's obj_mask, and skimage's slic labels (the 0.18.3 call, and the pre-processing of releases >= 0.19 assembled from 0.18.3's own
building blocks), its rgb2lab, and the label2rgb(kind='avg') image.  Writes tests/golden/skimage.npz.
"""
import os
import sys
import types

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"


def images():
    """uint8 RGB test images: objects of several shapes on dark / textured backgrounds, noise, a low-contrast one."""
    rng = np.random.RandomState(7)
    out = []
    for k, size in enumerate((96, 96, 128, 128, 160, 96, 112, 128)):
        yy, xx = np.mgrid[0:size, 0:size]
        base = rng.randint(60, 200, (1, 1, 3))
        img = np.clip(base + 25 * np.sin(xx[..., None] / (5.0 + k) + k) + rng.randint(-10, 10, (size, size, 3)), 0, 255)
        cy, cx = size * (0.45 + 0.02 * k), size * (0.5 - 0.015 * k)
        if k % 4 == 0:
            obj = ((yy - cy) ** 2 + (xx - cx) ** 2) < (size * 0.36) ** 2
        elif k % 4 == 1:
            obj = (((yy - cy) / (size * 0.40)) ** 2 + ((xx - cx) / (size * 0.22)) ** 2) < 1.0
        elif k % 4 == 2:
            obj = (np.abs(yy - cy) < size * 0.3) & (np.abs(xx - cx) < size * 0.18)
        else:
            obj = (np.abs(yy - cy) + np.abs(xx - cx)) < size * 0.42
        bg = 15 if k != 5 else base * 0.8 + rng.randint(-6, 6, (size, size, 3))          # k == 5: low contrast
        img = np.where(obj[..., None], img, bg)
        if k == 6:                                                                        # a hole and a second blob
            img = np.where((((yy - cy) ** 2 + (xx - cx) ** 2) < (size * 0.1) ** 2)[..., None], 15, img)
            img = np.where((((yy - size * 0.12) ** 2 + (xx - size * 0.85) ** 2) < (size * 0.06) ** 2)[..., None], 220, img)
        out.append(np.clip(img, 0, 255).astype(np.uint8))
    return out


def slic_rescaled(im, n_segments=5, sigma=2, compactness=10.0, max_iter=10):
    from scipy import ndimage as ndi
    from skimage.color import rgb2lab
    from skimage.segmentation._slic import _slic_cython, _enforce_label_connectivity_cython
    from skimage.segmentation.slic_superpixels import _get_grid_centroids
    image = im.astype(np.float64) / 255.0                       # img_as_float
    imin, imax = image.min(), image.max()
    image = image - imin
    if imax != imin:
        image = image / (imax - imin)
    image = rgb2lab(image[np.newaxis, ...])
    centroids, steps = _get_grid_centroids(image, n_segments)
    image = ndi.gaussian_filter(image, [sigma, sigma, sigma, 0])
    segments = np.ascontiguousarray(np.concatenate([centroids, np.zeros((centroids.shape[0], image.shape[3]))], axis=-1), dtype=np.float64)
    step = max(steps)
    image = np.ascontiguousarray(image * (1.0 / compactness), dtype=np.float64)
    labels = _slic_cython(image, None, segments, step, max_iter, np.ones(3), False, ignore_color=False, start_label=1)
    seg_size = np.prod(image.shape[:3]) / centroids.shape[0]
    labels = _enforce_label_connectivity_cython(labels, int(0.5 * seg_size), int(3 * seg_size), start_label=1)
    return np.asarray(labels)[0]


def main():
    import skimage
    from skimage import feature, color
    from skimage.segmentation import slic
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.ColorJitter = object
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tv.transforms
    sys.path.insert(0, REF_SRC)
    from self_supervised import dataset_generator as ref             # the REFERENCE's module
    assert ref.__file__.startswith(REF_SRC)
    out = {"versions": np.array([f"scikit-image {skimage.__version__}", f"numpy {np.__version__}",
                                 f"scipy {__import__('scipy').__version__}", f"Pillow {__import__('PIL').__version__}"])}
    imgs = images()
    out["n"] = np.int64(len(imgs))
    for i, im in enumerate(imgs):
        pil = Image.fromarray(im)
        gray = np.array(pil.convert("L"))
        out[f"img{i}"] = im
        out[f"gray{i}"] = gray
        out[f"canny{i}"] = feature.canny(gray, sigma=1.5, low_threshold=5, high_threshold=15)
        out[f"mask{i}"] = np.array(ref.obj_mask(pil).convert("1"))
        # (a) the 0.18.3 public call, labels from 1
        seg = slic(im, n_segments=5, sigma=2, convert2lab=True, start_label=1)
        out[f"slic18_{i}"] = seg.astype(np.int32)
        # (b) releases >= 0.19 stretch the float image to [0, 1] by its min / max first (and number labels from 1): the same
        # pipeline assembled from 0.18.3's own building blocks, in the order slic() calls them
        out[f"slic19_{i}"] = slic_rescaled(im).astype(np.int32)
        out[f"lab{i}"] = color.rgb2lab(im.astype(np.float64) / 255.0)
        # label2rgb(kind='avg') of >= 0.19 writes the float64 means into an array of the image's dtype
        avg = np.zeros_like(im)
        for lab_id in np.unique(out[f"slic19_{i}"]):
            m = (out[f"slic19_{i}"] == lab_id).nonzero()
            avg[m] = im[m].mean(axis=0)
        out[f"avg19_{i}"] = avg
        out[f"avg18f_{i}"] = np.asarray(color.label2rgb(seg, im, kind="avg"))       # 0.18.3: float64 output
    np.savez_compressed(os.path.join(HERE, "skimage.npz"), **out)
    print("skimage.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith(("canny", "mask"))})
    print([int(out[f"canny{i}"].sum()) for i in range(len(imgs))], [int(out[f"mask{i}"].sum()) for i in range(len(imgs))],
          [int(out[f"slic18_{i}"].max()) for i in range(len(imgs))], [int(out[f"slic19_{i}"].max()) for i in range(len(imgs))])


if __name__ == "__main__":
    main()
