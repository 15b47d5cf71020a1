#!/usr/bin/env python3
"""Column-reduction variants of the BatchNorm passes at the training shapes: which are byte-bound, which instruction-bound.
   python tools/colreduce_probe.py  ->  us and TB/s (operand bytes) per variant, fp32 and half tensors"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for dt in (torch.float32, torch.float16):
    for (n, h, c) in ((256, 64, 64), (256, 32, 128), (256, 16, 256), (256, 8, 512)):
        r = n * h * h
        es = 4 if dt == torch.float32 else 2
        dy = torch.randn(r, c, device=dev).to(dt); z = torch.randn(r, c, device=dev).to(dt)
        y = torch.randn(r, c, device=dev).to(dt)
        mask = torch.randint(0, 16, (r * c // 4,), device=dev, dtype=torch.uint8)
        mean, invstd = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        db, dg = torch.empty(c, device=dev), torch.empty(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        rows = [("stats(z)", lambda: ops.bn_stats(z.view(n, h, h, c), c, 1e-5, 0.1, rm, rv), es * r * c),
                ("reduce(dy, y, z)", lambda: ops.bn_bwd_reduce(dy, y, z, mean, invstd, db, dg, c), 3 * es * r * c),
                ("reduce zmask(dy, z)", lambda: ops.bn_bwd_zmask(dy.view(n, h, h, c), z.view(n, h, h, c), mean, invstd, gamma, beta, db, dg), None),
                ("reduce+apply mask(dy, m, z)", lambda: ops.bn_bwd_mask(dy, mask, z, mean, invstd, gamma, db, dg), None)]
        for name, fn, nb in rows:
            us = timed(fn)
            print(f"{str(dt)[6:]:8s} R={r:8d} C={c:4d} {name:30s} {us:8.1f} us" + (f"  {nb / us / 1e6:6.2f} TB/s" if nb else ""), flush=True)
