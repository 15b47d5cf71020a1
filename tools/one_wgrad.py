#!/usr/bin/env python3
"""Run one training-conv op a few times (for rocprofv3 --pmc runs).
usage: one_wgrad.py op(wgrad|dgrad|fwd|fwd1|fwd3|fwd6) N H Cin Cout k s p [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops
op = sys.argv[1]
n, h, cin, cout, k, s, p = map(int, sys.argv[2:9])
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 3
dev = torch.device("cuda:0")
x = torch.randn(n, h, h, cin, device=dev)
w = torch.randn(cout, k, k, cin, device=dev) * 0.05
ho = (h + 2 * p - k) // s + 1
dy = torch.randn(n, ho, ho, cout, device=dev)
dw = torch.empty(w.numel(), device=dev)
wft = ops.flip_transpose_weight(w)
for _ in range(iters):
    if op == "wgrad":
        ops.conv_wgrad(dy, x, dw, k, k, s, p)
    elif op == "dgrad":
        ops.conv_dgrad(dy, wft, x.shape, s, p)
    elif op in ("fwd3", "fwd6", "fwd1"):
        ops.conv_fwd(x, w, None, None, None, False, s, p, {"fwd3": 3, "fwd6": 6, "fwd1": True}[op])
    else:
        ops.conv_fwd(x, w, None, None, None, False, s, p)
torch.cuda.synchronize()
print("done")
