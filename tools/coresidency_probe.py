"""Do the input-gradient and the weight-gradient kernel of one conv layer run faster SIDE BY SIDE than one after the other?

At batch 32 both have about one workgroup per CU (profiles/r04_b32_trace_step.csv: ~80 us each for 61 us of matrix-pipe work).  Two
streams, no graph, no join inside the timed loop: stream A launches `reps` input gradients, stream B `reps` weight gradients
(+ their slab reductions); against the same launches alternating on one stream.  If (two streams) is not clearly below (one stream),
a horizontally fused dgrad + wgrad launch cannot pay either -- hipGraph fork/join cost is not in this measurement.
   python tools/coresidency_probe.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd")):
    sys.path.insert(0, p)
import torch
from self_supervised import ops

LAYERS = [  # name, H (input = output, stride 1), Cin, Cout
    ("layer1 3x3 64->64 @64", 64, 64, 64),
    ("layer2 3x3 128->128 @32", 32, 128, 128),
    ("layer3 3x3 256->256 @16", 16, 256, 256),
    ("layer4 3x3 512->512 @8", 8, 512, 512),
]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda", 0)
    reps = 40
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, h, cin, cout in LAYERS:
        x = torch.randn((n, h, h, cin), generator=g).to(dev)
        dy = torch.randn((n, h, h, cout), generator=g).to(dev)
        w = torch.randn((cout, 3, 3, cin), generator=g).to(dev) * 0.05
        wf = ops.flip_transpose_weight(w)
        dw = torch.empty(cout * 9 * cin, device=dev)
        c64 = cin == 64 and cout == 64

        def dgrad():
            if c64:
                return ops.conv3x3_c64(dy, wf)      # layer1's input gradient is the halo-tile kernel on the flipped filter
            return ops.conv_dgrad(dy, wf, x.shape, 1, 1)

        def wgrad():
            return ops.conv_wgrad(dy, x, dw, 3, 3, 1, 1)

        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e6 / reps

        def only_d():
            for _ in range(reps):
                dgrad()

        def only_w():
            for _ in range(reps):
                wgrad()

        def serial():
            for _ in range(reps):
                dgrad(); wgrad()

        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

        def side_by_side():
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur); sb.wait_stream(cur)
            for _ in range(reps):
                with torch.cuda.stream(sa):
                    dgrad()
                with torch.cuda.stream(sb):
                    wgrad()
            cur.wait_stream(sa); cur.wait_stream(sb)

        td, tw, ts, tp = timed(only_d), timed(only_w), timed(serial), timed(side_by_side)
        fl = 2.0 * n * h * h * cin * cout * 9
        print(f"{name:28s} batch {n}: dgrad {td:6.1f} us  wgrad+reduce {tw:6.1f} us  one stream {ts:6.1f} us  two streams {tp:6.1f} us   "
              f"(matrix pipe at peak: {2 * fl / 157.3e6:6.1f} us for both)", flush=True)


if __name__ == "__main__":
    main()
