"""Time of the fused cosine k-NN kernel (csrc/knn.hip) at the three WideResNet-50 scales of BASELINE configs[3] (64 images of 512 x 512:
128^2 / 64^2 / 32^2 positions with 256 / 512 / 1024 channels against a 588-row bank) and at the ResNet-18 scoring size.
   python tools/knn_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd")):
    sys.path.insert(0, p)
import torch
from self_supervised import ops

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for name, n, d, r in [("wrn50 layer1", 64 * 128 * 128, 256, 588), ("wrn50 layer2", 64 * 64 * 64, 512, 588), ("wrn50 layer3", 64 * 32 * 32, 1024, 588),
                      ("resnet18 patches", 256 * 841, 512, 588)]:
    x = torch.randn((n, d), generator=g).to(dev)
    bank = ops.l2_normalize_rows(torch.randn((r, d), generator=g).to(dev))
    for _ in range(2):
        out = ops.cosine_knn_fused(x, bank, 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        out = ops.cosine_knn_fused(x, bank, 3)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name:18s} N {n:8d} D {d:5d} R {r}: {dt * 1e3:7.3f} ms  {2.0 * n * d * r / dt / 1e12:6.1f} TFLOP/s  checksum {float(out.double().sum()):.9f}", flush=True)
