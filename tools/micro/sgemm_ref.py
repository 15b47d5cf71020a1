#!/usr/bin/env python3
"""Calibration: what the vendor library's fp32 GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this chip."""
import torch
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
for n in (4096, 8192, 16384):
    a = torch.randn(n, n, device=dev); b = torch.randn(n, n, device=dev)
    for _ in range(2):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 5 if n < 16384 else 2
    e0.record()
    for _ in range(it):
        c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"torch fp32 matmul {n}^3: {ms:.3f} ms  {2.0 * n ** 3 / ms / 1e9:.1f} TFLOP/s", flush=True)
# long back-to-back run (power controller settled): 8192^3, 40 launches, randn and zero operands
for name, mk in (("randn", torch.randn), ("zeros", torch.zeros)):
    a = mk(8192, 8192, device=dev); b = mk(8192, 8192, device=dev)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    print(f"torch fp32 matmul 8192^3 x40 {name}: {ms:.3f} ms  {2.0 * 8192 ** 3 / ms / 1e9:.1f} TFLOP/s", flush=True)
