#!/usr/bin/env python3
"""Times the BatchNorm-backward column reduction (ssad_bn_bwd_reduce_mask) and apply at the training shapes.  usage: bn_reduce_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops, _hip
dev = torch.device("cuda:0")
for (n, h, c) in ((256, 64, 64), (256, 32, 128), (256, 16, 256), (256, 8, 512), (32, 64, 64), (32, 8, 512)):
    r = n * h * h
    dy = torch.randn(r, c, device=dev); z = torch.randn(r, c, device=dev)
    mask = torch.randint(0, 16, (r * c // 4,), device=dev, dtype=torch.uint8)
    mean, invstd = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    db, dg = torch.empty(c, device=dev), torch.empty(c, device=dev)
    ws = torch.empty(_hip.lib().ssad_colreduce_workspace(r, c), device=dev, dtype=torch.float64)
    lib = _hip.lib()
    def red():
        _hip.check(lib.ssad_bn_bwd_reduce_mask(_hip.ptr(dy), mask.data_ptr(), _hip.ptr(z), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(db), _hip.ptr(dg), r, c, ws.data_ptr(), _hip.stream()))
    dz = torch.empty_like(dy)
    def app():
        _hip.check(lib.ssad_bn_apply_bwd_mask(_hip.ptr(dy), mask.data_ptr(), _hip.ptr(z), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(mean), _hip.ptr(db), _hip.ptr(dg), _hip.ptr(dz), r, c, _hip.stream()))
    for name, fn, nbytes in (("reduce", red, 4 * r * c * 2 + r * c // 4), ("apply", app, 4 * r * c * 3 + r * c // 4)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:6s} R={r:8d} C={c:4d}: {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)
