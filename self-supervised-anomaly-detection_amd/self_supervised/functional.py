"""Patch-window and prediction helpers (mirrors src/self_supervised/functional.py:27-29, :71-82)."""
import torch
from torch import Tensor


def get_prediction_class(predictions: Tensor) -> Tensor:
    return torch.max(predictions.data, 1).indices


def extract_patches(image: Tensor, dim: int = 32, stride: int = 4) -> Tensor:
    """(B,C,H,W) -> (B,P,C,dim,dim) zero-copy strided view; patch p = r*ncols + c covers rows
    [stride*r, +dim) and cols [stride*c, +dim).  PeraNet.forward never materialises this: the stem
    kernel applies the same window arithmetic in its loader (csrc/stem.hip)."""
    b, c, h, w = image.shape
    nr, nc = (h - dim) // stride + 1, (w - dim) // stride + 1
    sb, sc, sh, sw = image.stride()
    v = image.as_strided((b, nr, nc, c, dim, dim), (sb, sh * stride, sw * stride, sc, sh, sw))
    return v.reshape(b, nr * nc, c, dim, dim)


def extract_mask_patches(image: Tensor, dim: int = 32, stride: int = 4) -> Tensor:
    p = extract_patches(image, dim, stride)
    return p.reshape(-1, 1, dim, dim)
