"""Per-layer time of the 16-bit-operand halo weight gradient (csrc/wgrad_halo16.hip) + its slab reduction at batch 256, the four
3x3 / stride-1 shapes of ResNet-18.   python tools/wgh16_probe.py   (round 4: 142-162 us; the one-tap-per-workgroup kernel: ~330)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd")):
    sys.path.insert(0, p)
import torch
from self_supervised import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for name, h, c in [("layer1", 64, 64), ("layer2", 32, 128), ("layer3", 16, 256), ("layer4", 8, 512)]:
    x = torch.randn((256, h, h, c), generator=g).to(dev); dy = torch.randn((256, h, h, c), generator=g).to(dev)
    dw = torch.empty(c * 9 * c, device=dev)
    for _ in range(3): ops.conv_wgrad(dy, x, dw, 3, 3, 1, 1, bf16=2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.conv_wgrad(dy, x, dw, 3, 3, 1, 1, bf16=2)
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) / 20 * 1e6, 1), "us (wgrad + reduce)", flush=True)
