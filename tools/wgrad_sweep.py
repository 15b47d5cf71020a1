#!/usr/bin/env python3
"""Sweep the wgrad split count per training shape (SSAD_WGRAD_SPLITS) and print time per candidate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops
dev = torch.device("cuda:0")
B = 256
SHAPES = [("l1 3x3", 64, 64, 64, 3, 1, 1), ("l2 3x3/2", 64, 64, 128, 3, 2, 1), ("l2 1x1/2", 64, 64, 128, 1, 2, 0),
          ("l2 3x3", 32, 128, 128, 3, 1, 1), ("l3 3x3/2", 32, 128, 256, 3, 2, 1), ("l3 1x1/2", 32, 128, 256, 1, 2, 0),
          ("l3 3x3", 16, 256, 256, 3, 1, 1), ("l4 3x3/2", 16, 256, 512, 3, 2, 1), ("l4 1x1/2", 16, 256, 512, 1, 2, 0),
          ("l4 3x3", 8, 512, 512, 3, 1, 1)]
CAND = [8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 280, 320, 384, 448, 568, 640]


def timeit(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, h, cin, cout, k, s, p in SHAPES:
    x = torch.randn(B, h, h, cin, device=dev)
    ho = (h + 2 * p - k) // s + 1
    dy = torch.randn(B, ho, ho, cout, device=dev)
    dw = torch.empty(cout * k * k * cin, device=dev)
    os.environ.pop("SSAD_WGRAD_SPLITS", None)
    base = timeit(lambda: ops.conv_wgrad(dy, x, dw, k, k, s, p))
    from self_supervised import _hip
    auto = _hip.lib().ssad_wgrad_splits(B * ho * ho, cin, cout, k, k)
    res = []
    for c in CAND:
        if c * 256 > B * ho * ho:
            continue
        os.environ["SSAD_WGRAD_SPLITS"] = str(c)
        res.append((timeit(lambda: ops.conv_wgrad(dy, x, dw, k, k, s, p)), c))
    os.environ.pop("SSAD_WGRAD_SPLITS", None)
    best = min(res)
    print(f"{name:10s} auto={auto:4d} {base:.3f} ms | best {best[1]:4d} {best[0]:.3f} ms | " + " ".join(f"{c}:{t:.3f}" for t, c in res), flush=True)
