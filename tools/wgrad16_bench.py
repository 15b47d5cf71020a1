"""Time the 3x3 weight gradients of the precision-16 trunk (half tensors) at batch B: csrc/wgrad16.hip.  python tools/wgrad16_bench.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd")]
import torch
from self_supervised import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (hw, cin, cout, s) in [(64, 64, 64, 1), (32, 128, 128, 1), (16, 256, 256, 1), (8, 512, 512, 1), (64, 64, 128, 2), (32, 128, 256, 2), (16, 256, 512, 2)]:
    x = torch.randn(B, hw, hw, cin, device=dev).half()
    ho = (hw - 1) // s + 1
    dy = torch.randn(B, ho, ho, cout, device=dev).half()
    dw = torch.empty(cout * 9 * cin, device=dev)
    t = timeit(lambda: ops.conv_wgrad(dy, x, dw, 3, 3, s, 1, bf16=2))
    gf = 2.0 * B * ho * ho * cin * cout * 9 / 1e9
    print(f"{hw}x{hw} {cin}->{cout} s{s}: {t:.1f} us (wgrad + reduce), {gf / t * 1e3:.0f} TFLOP/s")
