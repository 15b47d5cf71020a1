"""Time the 3x3 / stride 1 convs of the precision-16 trunk at batch B on one GPU: conv16 halo kernel against the implicit GEMM / c64
forms over the same half tensors.  python tools/conv16_bench.py [B]"""
import os, sys, time
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


for (hw, c) in [(64, 64), (32, 128), (16, 256), (8, 512)]:
    x = torch.randn(B, hw, hw, c, device=dev).half()
    w = (torch.randn(c, 3, 3, c, device=dev) / (9 * c) ** 0.5).half()
    res = torch.randn(B, hw, hw, c, device=dev).half()
    tr = tuple(torch.rand(c, device=dev) + 0.5 for _ in range(4))
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    st = (1e-5, 0.1, rm, rv)
    gf = 2.0 * B * hw * hw * c * c * 9 / 1e9
    mb = 2.0 * (2 * x.numel() + w.numel()) / 1e6
    t = {}
    t["h plain"] = timeit(lambda: ops.conv3x3_h(x, w))
    t["h stats"] = timeit(lambda: ops.conv3x3_h(x, w, stats=st))
    t["h tr+emit+stats"] = timeit(lambda: ops.conv3x3_h(x, w, transform=tr, emit=True, stats=st))
    t["h residual"] = timeit(lambda: ops.conv3x3_h(x, w, residual=res))
    if ops.conv3x3_hw_ok(B, hw, hw, c, c):
        wp, _ = ops.conv3x3_hw_pack(w.float().reshape(-1), [(0, c, c, False)])
        t["hw plain"] = timeit(lambda: ops.conv3x3_hw(x, wp, c))
        t["hw stats"] = timeit(lambda: ops.conv3x3_hw(x, wp, c, stats=st))
        t["hw tr+emit+stats"] = timeit(lambda: ops.conv3x3_hw(x, wp, c, transform=tr, emit=True, stats=st))
        t["hw residual"] = timeit(lambda: ops.conv3x3_hw(x, wp, c, residual=res))
    if not os.environ.get("NO_IGEMM"):
        t["igemm stats"] = timeit(lambda: ops.conv_fwd_stats(x, w, 1e-5, 0.1, rm, rv, 1, 1, bf16=2))
    if c == 64 and not os.environ.get("NO_IGEMM"):
        wf = w.float()
        t["c64_h stats"] = timeit(lambda: ops.conv3x3_c64(x, wf, stats=st, bf16=2))
    print(f"{hw}x{hw}x{c}: {gf:.1f} GFLOP, {mb:.0f} MB in+out | " + " | ".join(f"{k} {v:.1f} us" for k, v in t.items()), flush=True)
