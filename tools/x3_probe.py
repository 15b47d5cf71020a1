#!/usr/bin/env python3
"""Accuracy (vs fp64) and speed of the exact-fp32 / bf16 / bf16x3 conv kernels on training and scoring shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
import torch.nn.functional as F
from self_supervised import ops
dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# accuracy on a small case against fp64
g = torch.Generator().manual_seed(0)
x = torch.randn(4, 12, 12, 128, generator=g); w = torch.randn(128, 3, 3, 128, generator=g) / (128 * 9) ** 0.5
ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), None, 1, 1).permute(0, 2, 3, 1)
for mode, name in ((False, "fp32"), (True, "bf16"), (3, "bf16x3"), (6, "bf16x6")):
    y = ops.conv_fwd(x.to(dev), w.to(dev), None, None, None, False, 1, 1, mode).cpu().double()
    print(f"accuracy {name:7s}: max |err| / max|ref| = {(y - ref).abs().max().item() / ref.abs().max().item():.3e}", flush=True)

B = 256
for name, h, cin, cout, k, s, p in [("l1 3x3", 64, 64, 64, 3, 1, 1), ("l2 3x3", 32, 128, 128, 3, 1, 1), ("l3 3x3", 16, 256, 256, 3, 1, 1),
                                     ("l4 3x3", 8, 512, 512, 3, 1, 1), ("l3 3x3/2", 32, 128, 256, 3, 2, 1)]:
    x = torch.randn(B, h, h, cin, device=dev); w = torch.randn(cout, k, k, cin, device=dev) * 0.05
    ho = (h + 2 * p - k) // s + 1
    dy = torch.randn(B, ho, ho, cout, device=dev); wft = ops.flip_transpose_weight(w)
    fl = 2.0 * B * ho * ho * cout * k * k * cin
    row = []
    for mode in (False, True, 3, 6):
        tf = timeit(lambda: ops.conv_fwd(x, w, None, None, None, False, s, p, mode))
        td = timeit(lambda: ops.conv_dgrad(dy, wft, x.shape, s, p, bf16=mode))
        row.append(f"{tf:.3f}/{td:.3f} ms ({fl / tf / 1e9:.0f}/{fl / td / 1e9:.0f} TF/s)")
    print(f"{name:9s} fwd/dgrad  fp32 {row[0]} | bf16 {row[1]} | x3 {row[2]} | x6 {row[3]}", flush=True)
# scoring layout
n = 15979
for name, h, c in [("s-l1", 16, 64), ("s-l2", 8, 128), ("s-l3", 4, 256), ("s-l4", 2, 512)]:
    x = torch.randn(h, h, n, c, device=dev); w = torch.randn(c, 3, 3, c, device=dev) * 0.05
    fl = 2.0 * n * h * h * c * 9 * c
    t0 = timeit(lambda: ops.conv_fwd_hwnc(x, w, None, None, None, True, 1, 1))
    t3 = timeit(lambda: ops.conv_fwd_hwnc(x, w, None, None, None, True, 1, 1, x3=True))
    t6 = timeit(lambda: ops.conv_fwd_hwnc(x, w, None, None, None, True, 1, 1, x3=6))
    print(f"{name:9s} hwnc fwd   fp32 {t0:.3f} ms ({fl / t0 / 1e9:.0f}) | x3 {t3:.3f} ms ({fl / t3 / 1e9:.0f}) | x6 {t6:.3f} ms ({fl / t6 / 1e9:.0f} TF/s alg)", flush=True)
