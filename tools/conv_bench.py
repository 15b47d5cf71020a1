#!/usr/bin/env python3
"""Per-layer micro-benchmark of the MFMA conv kernels (forward / dgrad / wgrad) at the shapes the two phases use.

    python tools/conv_bench.py [score|train|all] [--iters 5]
Prints TFLOP/s per shape (algorithmic FLOPs, HIP-event timing on the launch stream)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch
from self_supervised import ops

# (name, N, H, W, Cin, Cout, k, stride, pad)
SCORE = [("l1 3x3 64>64 16x16", 8192, 16, 16, 64, 64, 3, 1, 1),
         ("l2 3x3s2 64>128", 8192, 16, 16, 64, 128, 3, 2, 1),
         ("l2 1x1s2 64>128", 8192, 16, 16, 64, 128, 1, 2, 0),
         ("l2 3x3 128>128 8x8", 8192, 8, 8, 128, 128, 3, 1, 1),
         ("l3 3x3s2 128>256", 8192, 8, 8, 128, 256, 3, 2, 1),
         ("l3 3x3 256>256 4x4", 8192, 4, 4, 256, 256, 3, 1, 1),
         ("l4 3x3s2 256>512", 8192, 4, 4, 256, 512, 3, 2, 1),
         ("l4 3x3 512>512 2x2", 8192, 2, 2, 512, 512, 3, 1, 1),
         ("head 896>512", 215296, 1, 1, 896, 512, 1, 1, 0),
         ("knn 512>588", 215296, 1, 1, 512, 588, 1, 1, 0)]
TRAIN = [("l1 3x3 64>64 64x64", 256, 64, 64, 64, 64, 3, 1, 1),
         ("l2 3x3s2 64>128", 256, 64, 64, 64, 128, 3, 2, 1),
         ("l2 3x3 128>128 32x32", 256, 32, 32, 128, 128, 3, 1, 1),
         ("l3 3x3 256>256 16x16", 256, 16, 16, 256, 256, 3, 1, 1),
         ("l4 3x3 512>512 8x8", 256, 8, 8, 512, 512, 3, 1, 1),
         ("stem im2col 160>64", 256, 128, 128, 160, 64, 1, 1, 0)]


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 5
    dev = torch.device("cuda:0")
    for tag, shapes in (("score", SCORE), ("train", TRAIN)):
        if which not in ("all", tag):
            continue
        for name, n, h, w, cin, cout, k, s, p in shapes:
            x = torch.randn(n, h, w, cin, device=dev)
            wt = torch.randn(cout, k, k, cin, device=dev) * 0.05
            sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
            ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
            flops = 2.0 * n * ho * wo * cout * k * k * cin
            t = timeit(lambda: ops.conv_fwd(x, wt, sc, sh, None, True, s, p), iters)
            line = f"{tag:5s} {name:24s} fwd {flops / t / 1e12:6.1f} TF/s ({t * 1e3:7.3f} ms)"
            if tag == "score" and h > 1:
                xh = x.permute(1, 2, 0, 3).contiguous()
                t = timeit(lambda: ops.conv_fwd_hwnc(xh, wt, sc, sh, None, True, s, p), iters)
                line += f" | hwnc {flops / t / 1e12:6.1f} ({t * 1e3:7.3f} ms)"
                if k == 3 and s == 1 and cin == 64 and cout == 64:
                    t = timeit(lambda: ops.conv3x3_c64_eval(xh, wt, sc, sh, None, True, True, True), iters)
                    line += f" | c64 hwnc {flops / t / 1e12:6.1f} ({t * 1e3:7.3f} ms)"
                    t = timeit(lambda: ops.conv3x3_c64_eval(x, wt, sc, sh, None, True, False, False), iters)
                    line += f" | c64 nhwc {flops / t / 1e12:6.1f} ({t * 1e3:7.3f} ms)"
            if tag == "train":
                dy = torch.randn(n, ho, wo, cout, device=dev)
                if cout % 32 == 0 and k > 0 and "im2col" not in name:
                    wf = ops.flip_transpose_weight(wt)
                    t = timeit(lambda: ops.conv_dgrad(dy, wf, x.shape, s, p), iters)
                    line += f" | dgrad {flops / t / 1e12:6.1f} ({t * 1e3:7.3f} ms)"
                dw = torch.empty(cout * k * k * cin, device=dev)
                t = timeit(lambda: ops.conv_wgrad(dy, x, dw, k, k, s, p), iters)
                line += f" | wgrad {flops / t / 1e12:6.1f} ({t * 1e3:7.3f} ms)"
            print(line, flush=True)


if __name__ == "__main__":
    main()
