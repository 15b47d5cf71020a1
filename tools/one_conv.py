#!/usr/bin/env python3
"""Run one conv shape a few times (for rocprofv3 --pmc runs).  usage: one_conv.py N H W Cin Cout k s p layout(nhwc|hwnc) [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops
n, h, w, cin, cout, k, s, p = map(int, sys.argv[1:9])
layout = sys.argv[9]
iters = int(sys.argv[10]) if len(sys.argv) > 10 else 3
dev = torch.device("cuda:0")
x = torch.randn((n, h, w, cin) if layout == "nhwc" else (h, w, n, cin), device=dev)
wt = torch.randn(cout, k, k, cin, device=dev) * 0.05
f = ops.conv_fwd if layout == "nhwc" else ops.conv_fwd_hwnc
for _ in range(iters):
    y = f(x, wt, None, None, None, True, s, p)
torch.cuda.synchronize()
print("done", tuple(y.shape))
