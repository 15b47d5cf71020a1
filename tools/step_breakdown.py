#!/usr/bin/env python3
"""Per-kernel time of one training step for a given precision (32 | 16 | bf16x3 | bf16x6): step_breakdown.py PREC [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops, training
from self_supervised.models import PeraNet
prec = sys.argv[1] if len(sys.argv) > 1 else "32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = PeraNet().to(dev).train(); m.unfreeze()
x = torch.randn(B, 3, 256, 256, device=dev); y = torch.randint(0, 4, (B,), device=dev)
st = training.DataParallelStep(m, lr=0.005, world_size=1, precision=int(prec) if prec.isdigit() else prec)
for _ in range(3):
    st.step(x, y)
torch.cuda.synchronize()
ops.PROFILE = []
N = 5
import time
t0 = time.perf_counter()
for _ in range(N):
    st.step(x, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
by = {}
for r in ops.drain_profile():
    e = by.setdefault(r["kernel"], [0.0, 0, 0.0]); e[0] += r["ms"]; e[1] += 1; e[2] += r["flops"]
print(f"precision {prec}: {dt * 1e3:.2f} ms/step ({B / dt:.0f} img/s)")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:18s} {v[0] / N:7.3f} ms  x{v[1] // N:3d}  {v[2] / max(v[0], 1e-9) / 1e9:6.1f} TF/s")
