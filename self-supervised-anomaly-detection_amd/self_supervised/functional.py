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


# ---- file-system helpers of the MVTec layout <root>/<category>/{train/good,test/<defect>,ground_truth/<defect>} ----
# (mirror src/self_supervised/functional.py:14-68)
import glob as _glob
import os as _os

import numpy as _np
from PIL import Image as _Image


def get_all_subject_experiments(dataset_dir: str):
    return sorted(d for d in _os.listdir(dataset_dir) if _os.path.isdir(_os.path.join(dataset_dir, d)))


def get_subdirectories(main_path: str):
    return _np.array(sorted(d for d in _os.listdir(main_path) if _os.path.isdir(_os.path.join(main_path, d))), dtype=str)


def get_filenames(main_path: str):
    return _np.array(sorted(f.replace("\\", '/') for f in _glob.glob(main_path + '*.png')))


def get_test_data_filenames(main_path: str):
    parts = [get_filenames(main_path + d + '/') for d in get_subdirectories(main_path)]
    return _np.concatenate(parts) if parts else _np.empty(0, dtype=str)


def get_ground_truth_filename(test_filename: str, ground_truth_dir: str):
    """.../test/<defect>/<id>.png -> <ground_truth_dir><defect>/<id>_mask.png; None for 'good' images."""
    head, defect, image_name = test_filename.rsplit('/', 2)
    if defect == 'good':
        return None
    stem, ext = image_name.split('.')[0], image_name.split('.')[1]
    return ground_truth_dir + defect + '/' + stem + '_mask.' + ext


def get_ground_truth(filename: str = None, imsize=(256, 256)):
    if filename:
        return _Image.open(filename).resize(imsize).convert('1')
    return _Image.new(mode='1', size=imsize)


def duplicate_filenames(filenames, baseline: int = 2000):
    """Repeat the whole list until it holds at least ``baseline`` names."""
    out = _np.array(filenames, copy=True)
    while out.shape[0] < baseline:
        out = _np.concatenate([out, filenames], dtype=str)
    return out
