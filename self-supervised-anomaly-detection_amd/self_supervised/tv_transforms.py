"""The torchvision.transforms the reference uses on PIL images (src/self_supervised/datasets.py:44-47, :102-105, :221,
:253, :391-394), restated from torchvision's public behaviour (v0.13+ semantics) because torchvision is not installed:
ToTensor, Normalize, Compose, ColorJitter, RandomAffine, RandomCrop.  THIRD-PARTY RESTATEMENT: parity of these classes is
against the documented algorithm, not against a run of torchvision.  What matters for the reference's random streams is
kept exactly: which torch RNG calls are made, in which order (ColorJitter: randperm(4) then one uniform_ per enabled
factor; RandomAffine: angle then scale; RandomCrop: row then column, and no draw at all when the crop is the image).

Stand-alone on purpose (no package-relative imports): tests/golden/make_fixtures.py loads this file as the
``torchvision.transforms`` stub when it imports the reference's datasets.py."""
import math

import numpy as np
import torch
from PIL import Image, ImageEnhance


def to_tensor(img):
    """PIL -> float32 CHW in [0,1] (mode '1' -> {0,1})."""
    a = np.array(img.convert('L') if img.mode == '1' else img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(a.transpose(2, 0, 1).copy()).float().div_(255.0)


class ToTensor:
    def __call__(self, img):
        return to_tensor(img)


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean).view(-1, 1, 1)
        self.std = torch.tensor(std).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


class Compose:
    def __init__(self, ts):
        self.ts = list(ts)

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class ColorJitter:
    """brightness / contrast / saturation factors ~ U(max(0, 1-d), 1+d), applied in a random order; hue disabled (0).
    torchvision draws ``randperm(4)`` (brightness, contrast, saturation, hue) and then the enabled factors in that fixed
    order; a disabled op keeps its slot in the permutation and is skipped."""
    _ENH = (ImageEnhance.Brightness, ImageEnhance.Contrast, ImageEnhance.Color)

    def __init__(self, brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0):
        assert hue == 0.0, "hue jitter is not used by the reference and not restated"
        self.b, self.c, self.s = brightness, contrast, saturation

    def sample(self):
        """(order of the three enabled ops, their factors) with torchvision's RNG consumption."""
        perm = torch.randperm(4).tolist()
        f = [float(torch.empty(1).uniform_(max(0.0, 1 - d), 1 + d)) if d else None for d in (self.b, self.c, self.s)]
        order = [op for op in perm if op < 3 and f[op] is not None]
        return order, [1.0 if v is None else v for v in f]

    def __call__(self, img):
        order, f = self.sample()
        for op in order:
            img = self._ENH[op](img).enhance(f[op])
        return img


def inverse_affine_matrix(center, angle, translate, scale):
    """torchvision.transforms.functional._get_inverse_affine_matrix with zero shear, same operation order."""
    rot = math.radians(angle)
    cx, cy = center
    tx, ty = translate
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / scale for x in m]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


class RandomAffine:
    """Rotation ~ U(-deg, deg) and zoom ~ U(scale) about the image centre, nearest resampling, zero fill."""

    def __init__(self, degrees, scale=None):
        self.deg, self.scale = float(degrees), scale

    def sample(self):
        ang = float(torch.empty(1).uniform_(-self.deg, self.deg).item())
        sc = float(torch.empty(1).uniform_(self.scale[0], self.scale[1]).item()) if self.scale is not None else 1.0
        return ang, sc

    def __call__(self, img):
        ang, sc = self.sample()
        w, h = img.size
        m = inverse_affine_matrix((w * 0.5, h * 0.5), ang, (0, 0), sc)
        return img.transform((w, h), Image.AFFINE, m, Image.NEAREST)


class RandomCrop:
    def __init__(self, size):
        self.size = size

    def sample(self, w, h):
        if w == self.size and h == self.size:
            return 0, 0                                      # torchvision returns early: no RNG draw
        top = int(torch.randint(0, h - self.size + 1, size=(1,)).item())
        left = int(torch.randint(0, w - self.size + 1, size=(1,)).item())
        return top, left

    def __call__(self, img):
        w, h = img.size
        top, left = self.sample(w, h)
        return img.crop((left, top, left + self.size, top + self.size))
