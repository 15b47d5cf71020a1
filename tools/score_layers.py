#!/usr/bin/env python3
"""Launch-by-launch time of one scoring chunk (19 images x 841 patches through the eval trunk): score_layers.py [images]
Prints every kernel launch of trunk_eval in order with its algorithmic and executed TFLOP/s (HIP events on the launch stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops
from self_supervised.models import PeraNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 19
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = PeraNet().to(dev).eval(); m.enable_patch_level_mode()
x = torch.rand(B, 3, 256, 256, device=dev)
for _ in range(2):
    m(x)
torch.cuda.synchronize()
R = 4
ops.PROFILE = []
for _ in range(R):
    m(x)
recs = ops.drain_profile()
ops.PROFILE = None
n = len(recs) // R
tot = 0.0
for i in range(n):
    ms = sorted(recs[i + k * n]["ms"] for k in range(R))[R // 2]
    r = recs[i]
    tot += ms
    print(f"{i:3d} {r['kernel']:18s} {ms:8.3f} ms  alg {r['flops'] / ms / 1e9:6.1f}  exec {r['exec_flops'] / ms / 1e9:6.1f} TF/s", flush=True)
print(f"total {tot:.3f} ms for {B} images = {B * 841} patches")
