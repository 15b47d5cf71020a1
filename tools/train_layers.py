#!/usr/bin/env python3
"""Per-shape timing of the training convolutions of ResNet-18 (batch 256 @ 256x256): fwd, dgrad, wgrad.
Usage: train_layers.py [batch] [bf16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BF = len(sys.argv) > 2 and sys.argv[2] == "bf16"
# (name, H, Cin, Cout, k, stride, pad, count_fwd, count_dgrad)
SHAPES = [("l1 3x3", 64, 64, 64, 3, 1, 1, 4, 4),
          ("l2 3x3/2", 64, 64, 128, 3, 2, 1, 1, 1), ("l2 1x1/2", 64, 64, 128, 1, 2, 0, 1, 1), ("l2 3x3", 32, 128, 128, 3, 1, 1, 3, 3),
          ("l3 3x3/2", 32, 128, 256, 3, 2, 1, 1, 1), ("l3 1x1/2", 32, 128, 256, 1, 2, 0, 1, 1), ("l3 3x3", 16, 256, 256, 3, 1, 1, 3, 3),
          ("l4 3x3/2", 16, 256, 512, 3, 2, 1, 1, 1), ("l4 1x1/2", 16, 256, 512, 1, 2, 0, 1, 1), ("l4 3x3", 8, 512, 512, 3, 1, 1, 3, 3)]


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


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print(f"{'shape':10s} {'fwd ms':>8s} {'TF/s':>6s} {'dgrad ms':>9s} {'TF/s':>6s} {'wgrad ms':>9s} {'TF/s':>6s}  x")
for name, h, cin, cout, k, s, p, nf, nd in SHAPES:
    x = torch.randn(B, h, h, cin, device=dev)
    w = torch.randn(cout, k, k, cin, device=dev) * 0.05
    ho = (h + 2 * p - k) // s + 1
    dy = torch.randn(B, ho, ho, cout, device=dev)
    wft = ops.flip_transpose_weight(w)
    dw = torch.empty(w.numel(), device=dev)
    fl = 2.0 * B * ho * ho * cout * k * k * cin
    tf = timeit(lambda: ops.conv_fwd_stats(x, w, 1e-5, 0.1, None, None, s, p, BF))
    tf0 = timeit(lambda: ops.conv_fwd(x, w, None, None, None, False, s, p, BF))
    td = timeit(lambda: ops.conv_dgrad(dy, wft, x.shape, s, p, bf16=BF))
    tw = timeit(lambda: ops.conv_wgrad(dy, x, dw, k, k, s, p, bf16=BF))
    os.environ["SSAD_WGRAD_HALO"] = "0"
    tw0 = timeit(lambda: ops.conv_wgrad(dy, x, dw, k, k, s, p, bf16=BF))
    os.environ["SSAD_WGRAD_HALO"] = "1"
    tot["fwd"] += nf * tf; tot["dgrad"] += nd * td; tot["wgrad"] += nf * tw
    print(f"{name:10s} {tf:8.3f} {fl / tf / 1e9:6.1f} {td:9.3f} {fl / td / 1e9:6.1f} {tw:9.3f} {fl / tw / 1e9:6.1f}  {nf}  (fwd without stats {tf0:.3f}; wgrad split-kernel {tw0:.3f})", flush=True)
print("totals (ms, with multiplicity):", {k: round(v, 2) for k, v in tot.items()})
