"""Image files -> uint8 RGB batches on the device at the working size.

The reference opens every file as ``Image.open(f).resize(imsize).convert('RGB')`` on the host (src/self_supervised/datasets.py:68,
:211-213).  Here only the PNG decode stays on the host (Pillow, in threads: it releases the GIL); the decoded pixels go to the
device at their NATIVE size and Pillow's bicubic resize runs there (csrc/resize.hip, bit-exact: tests/test_hip_resize.py), then the
'L' -> 'RGB' replication.  Modes the kernel does not cover (palette, alpha, 16-bit) take Pillow's own resize on the host."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from PIL import Image

from . import ops

_GPU_MODES = ("L", "RGB")


def read_native(filename):
    """The decoded file as (mode, H x W [x C] uint8 array) when the device resize covers its mode, else the PIL image itself."""
    img = Image.open(filename)
    if img.mode in _GPU_MODES:
        return img.mode, np.asarray(img)
    img.load()
    return None, img


def to_rgb_batch(items, size, device, chunk=32):
    """items: what read_native returned, in order; size = (width, height) as PIL takes it.  -> uint8 [n][h][w][3] on `device`,
    equal to np.asarray(Image.open(f).resize(size).convert('RGB')) for every file."""
    w, h = int(size[0]), int(size[1])
    n = len(items)
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=device)
    groups = {}
    for i, (mode, a) in enumerate(items):
        if mode is None:                                         # Pillow's own path (rare modes)
            out[i] = torch.from_numpy(np.array(a.resize((w, h)).convert('RGB'))).to(device)
        else:
            groups.setdefault((mode, a.shape[0], a.shape[1]), []).append(i)
    for (mode, hin, win), idx in groups.items():
        for s in range(0, len(idx), chunk):
            part = idx[s:s + chunk]
            host = np.stack([items[i][1] for i in part])
            if host.ndim == 3:
                host = host[..., None]
            dev = torch.from_numpy(np.ascontiguousarray(host)).to(device)
            dev = ops.resize_bicubic_u8(dev, (w, h))                 # the input itself when the size already matches
            if dev.shape[-1] == 1:
                dev = dev.expand(-1, -1, -1, 3)                      # Convert.c l2rgb: the grey value in all three channels
            out[torch.as_tensor(part, device=device)] = dev
    return out


def load_rgb_batch(filenames, size, device, threads=8):
    """np.asarray(Image.open(f).resize(size).convert('RGB')) for every file, as one uint8 device batch."""
    with ThreadPoolExecutor(max(1, min(threads, len(filenames)))) as pool:
        items = list(pool.map(read_native, filenames))
    return to_rgb_batch(items, size, device)
