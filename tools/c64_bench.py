#!/usr/bin/env python3
"""Layer1 conv (64 -> 64, 3x3, 64x64 maps): halo-tile kernel vs the implicit-GEMM kernel.  c64_bench.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


x = torch.randn(B, H, H, 64, device=dev)
w = torch.randn(64, 3, 3, 64, device=dev) * 0.05
res = torch.randn(B, H, H, 64, device=dev)
mean, invstd, gamma, beta = (torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5, torch.rand(64, device=dev) + 0.5,
                             torch.randn(64, device=dev) * 0.1)
fl = 2.0 * B * H * H * 64 * 9 * 64
rows = [("igemm plain", lambda: ops.conv_fwd(x, w, None, None, None, False, 1, 1)),
        ("igemm + stats", lambda: ops.conv_fwd_stats(x, w, 1e-5, 0.1, None, None, 1, 1)),
        ("igemm dgrad + residual", lambda: ops.conv_dgrad(x, w, x.shape, 1, 1, res)),
        ("halo plain", lambda: ops.conv3x3_c64(x, w)),
        ("halo + residual", lambda: ops.conv3x3_c64(x, w, residual=res)),
        ("halo + stats", lambda: ops.conv3x3_c64(x, w, stats=(1e-5, 0.1, None, None))),
        ("halo + transform + stats", lambda: ops.conv3x3_c64(x, w, transform=(mean, invstd, gamma, beta), stats=(1e-5, 0.1, None, None))),
        ("halo + transform + emit + stats", lambda: ops.conv3x3_c64(x, w, transform=(mean, invstd, gamma, beta), emit=True, stats=(1e-5, 0.1, None, None))),
        ("bn_apply_fwd alone", lambda: ops.bn_apply_fwd(x, mean, invstd, gamma, beta, None, True))]
for name, fn in rows:
    t = timeit(fn)
    print(f"{name:34s} {t:7.3f} ms  {fl / t / 1e9:6.1f} TF/s", flush=True)
