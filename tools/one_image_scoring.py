"""Time of the patch-level forward of ONE 256 x 256 image (841 patches), the unit tools.inference works in (batch size 1):
   python tools/one_image_scoring.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("SSAD_ALLOW_RANDOM_BACKBONE", "1")
import torch
from self_supervised import ops
from self_supervised.models import PeraNet

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
m = PeraNet().to(dev).eval()
m.enable_patch_level_mode()
x = torch.randn(b, 3, 256, 256, device=dev)
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = m(x)
    torch.cuda.synchronize()
    print("ms per forward of", b, "image(s):", (time.perf_counter() - t0) * 100)
    ops.PROFILE = []
    m(x)
    recs = ops.drain_profile()
    ops.PROFILE = None
agg = {}
for r in recs:
    a = agg.setdefault(r["kernel"], [0, 0.0])
    a[0] += 1
    a[1] += r["ms"]
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:28s} {n:4d} {ms:9.3f} ms")
print("sum", sum(r["ms"] for r in recs))
