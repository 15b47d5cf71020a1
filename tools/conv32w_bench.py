"""Time the 3x3 / stride 1 convs of the exact-fp32 trunk at batch B on one GPU: the register-fed form (csrc/conv16w.hip, T = float)
against the kernels it replaces (halo c64 kernel on layer1, implicit GEMM elsewhere).  python tools/conv32w_bench.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd")]
import torch
from self_supervised import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)


def timeit(fn, n=10):
    for _ in range(2):
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
    x = torch.randn(B, hw, hw, c, device=dev)
    w = torch.randn(c, 3, 3, c, device=dev) / (9 * c) ** 0.5
    res = torch.randn(B, hw, hw, c, device=dev)
    tr = tuple(torch.rand(c, device=dev) + 0.5 for _ in range(4))
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    st = (1e-5, 0.1, rm, rv)
    gf = 2.0 * B * hw * hw * c * c * 9 / 1e9
    t = {}
    if ops.conv3x3_hw_ok(B, hw, hw, c, c, f32=True):
        wp, _ = ops.conv3x3_hw_pack(w.reshape(-1), [(0, c, c, False)], f32=True)
        t["fw plain"] = timeit(lambda: ops.conv3x3_hw(x, wp, c))
        t["fw stats"] = timeit(lambda: ops.conv3x3_hw(x, wp, c, stats=st))
        t["fw tr+emit+stats"] = timeit(lambda: ops.conv3x3_hw(x, wp, c, transform=tr, emit=True, stats=st))
        t["fw residual"] = timeit(lambda: ops.conv3x3_hw(x, wp, c, residual=res))
    t["igemm stats"] = timeit(lambda: ops.conv_fwd_stats(x, w, 1e-5, 0.1, rm, rv, 1, 1))
    wf = ops.flip_transpose_weight(w)
    t["igemm dgrad+res"] = timeit(lambda: ops.conv_dgrad(res, wf, tuple(x.shape), 1, 1, residual=x))
    if c == 64:
        t["c64 stats"] = timeit(lambda: ops.conv3x3_c64(x, w, stats=st))
        t["c64 tr+emit+stats"] = timeit(lambda: ops.conv3x3_c64(x, w, transform=tr, emit=True, stats=st))
    print(f"{hw}x{hw}x{c}: {gf:.1f} GFLOP | " + " | ".join(f"{k} {v:.0f} us ({gf / v * 1e3:.0f} TFLOP/s)" for k, v in t.items()), flush=True)
