"""Builds a tiny MVTec-AD-shaped tree of synthetic PNGs (the real dataset is not available offline)."""
import os

import numpy as np
from PIL import Image


def make_tree(root, categories=("bottle", "carpet"), n_train=10, n_test_good=3, n_test_bad=3, size=96, seed=0):
    rng = np.random.RandomState(seed)
    for cat in categories:
        base = rng.randint(40, 200, (1, 1, 3))
        for split, n in (("train/good", n_train), ("test/good", n_test_good), ("test/broken", n_test_bad)):
            d = os.path.join(root, cat, split)
            os.makedirs(d, exist_ok=True)
            for i in range(n):
                yy, xx = np.mgrid[0:size, 0:size]
                img = np.clip(base + 30 * np.sin(xx[..., None] / 7.0 + i) + rng.randint(-12, 12, (size, size, 3)), 0, 255)
                blob = ((yy - size / 2) ** 2 + (xx - size / 2) ** 2) < (size * 0.38) ** 2
                img = np.where(blob[..., None], img, 15 if cat != "carpet" else img)
                gt = np.zeros((size, size), np.uint8)
                if split == "test/broken":
                    y0, x0 = rng.randint(size // 4, size // 2, 2)
                    img[y0:y0 + size // 6, x0:x0 + size // 5] = 255 - img[y0:y0 + size // 6, x0:x0 + size // 5]
                    gt[y0:y0 + size // 6, x0:x0 + size // 5] = 255
                    g = os.path.join(root, cat, "ground_truth/broken")
                    os.makedirs(g, exist_ok=True)
                    Image.fromarray(gt).save(os.path.join(g, f"{i:03d}_mask.png"))
                Image.fromarray(img.astype(np.uint8)).save(os.path.join(d, f"{i:03d}.png"))
    return root + "/" if not root.endswith("/") else root
