#!/usr/bin/env python3
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for dbg in (0, 7, 15, 31, 23, 8):
        env = dict(os.environ, SSAD_WGRAD_DEBUG=str(dbg))
        out = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True).stdout.strip()
        print(f"debug={dbg}: {out}", flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops, _hip
dev = torch.device("cuda:0")
res = []
for (n, h, cin, cout, splits) in [(256, 32, 128, 128, 56), (256, 32, 128, 128, 224), (256, 8, 512, 512, 16)]:
    x = torch.randn(n, h, h, cin, device=dev); dy = torch.randn(n, h, h, cout, device=dev)
    m = dy.numel() // cout
    slab = torch.empty(splits, cout, 9 * cin, device=dev)
    f = lambda: _hip.lib().ssad_conv_wgrad(_hip.ptr(dy), _hip.ptr(x), _hip.ptr(slab), splits, n, h, h, cin, cout, 3, 3, 1, 1, _hip.stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    res.append(f"{t:.3f} ms {2.0 * m * cout * 9 * cin / t / 1e9:.0f} TF/s")
print(" | ".join(res))
