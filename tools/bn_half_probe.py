#!/usr/bin/env python3
"""Times the half-tensor BatchNorm kernels of the precision-16 step at the training shapes (batch 256): reductions (mask from the stored
activation / recomputed from z), backward apply, forward apply.  usage: bn_half_probe.py [batch, default 256] [f32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F32 = len(sys.argv) > 2 and sys.argv[2] == "f32"
for (n, h, c) in ((B, 64, 64), (B, 32, 128), (B, 16, 256), (B, 8, 512)):
    shape = (n, h, h, c)
    dy, z, ya, res = ((torch.randn(shape, device=dev) if F32 else torch.randn(shape, device=dev).half()) for _ in range(4))
    mean, invstd, gamma, beta = (torch.rand(c, device=dev) + 0.5 for _ in range(4))
    db, dg = torch.empty(c, device=dev), torch.empty(c, device=dev)
    e = dy.numel() * dy.element_size()
    rows = [("reduce(dy, yact, z)", lambda: ops.bn_bwd_reduce(dy, ya, z, mean, invstd, db, dg, c), 3 * e),
            ("apply_bwd(dy, yact, z) + dres", lambda: ops.bn_apply_bwd(dy, ya, z, mean, invstd, gamma, db, dg, True), 5 * e),
            ("reduce + apply zmask(dy, z)", lambda: ops.bn_bwd_zmask(dy, z, mean, invstd, gamma, beta, db, dg), 5 * e),
            ("apply_fwd(z, res)", lambda: ops.bn_apply_fwd(z, mean, invstd, gamma, beta, res, True), 3 * e)]
    for name, fn, nbytes in rows:
        ms = timeit(fn)
        print(f"{str(shape):22s} {name:32s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:7.2f} TB/s", flush=True)
