"""Bit-for-bit comparison of ssad_conv3x3_fw between two builds of the library (ab_tmp/lib_old.so, ab_tmp/lib_new.so): outputs, emitted
activation, BatchNorm statistics.  python tools/micro/ab_bits_conv32w.py"""
import ctypes as C
import sys
import torch

libs = [C.CDLL("ab_tmp/lib_old.so"), C.CDLL("ab_tmp/lib_new.so")]
for L in libs:
    L.ssad_conv3x3_hw_stats_rows.restype = C.c_int64
    L.ssad_conv3x3_hw_stats_rows.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int]
    L.ssad_last_error.restype = C.c_char_p
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def run(L, x, w, N, H, W, Cc, tr, res, mask):
    packed = torch.empty(Cc * 9 * Cc, device=dev)
    desc = (C.c_int64 * 5)(0, 0, Cc, Cc, 0)
    assert L.ssad_conv3x3_fw_pack_batch(P(w), P(packed), desc, 1, C.c_void_p(0)) == 0
    out = torch.empty_like(x)
    emit = torch.zeros_like(x) if tr is not None else None
    rows = L.ssad_conv3x3_hw_stats_rows(C.c_int64(N), H, W, Cc)
    ws = torch.zeros(rows * 2 * Cc, dtype=torch.float64, device=dev)
    mean = torch.empty(Cc, device=dev); invstd = torch.empty(Cc, device=dev)
    rm = torch.zeros(Cc, device=dev); rv = torch.ones(Cc, device=dev)
    t = tr if tr is not None else [None] * 4
    rc = L.ssad_conv3x3_fw(P(x), P(packed), P(out), P(res), P(mask), P(t[0]), P(t[1]), P(t[2]), P(t[3]), P(emit), C.c_int64(N), H, W, Cc, Cc,
                           P(ws), C.c_float(1e-5), C.c_float(0.1), P(mean), P(invstd), P(rm), P(rv), C.c_void_p(0))
    assert rc == 0, L.ssad_last_error()
    torch.cuda.synchronize()
    return out, emit, mean, invstd, rm, rv


bad = 0
for (N, H, W, Cc) in [(96, 64, 64, 64), (7, 64, 64, 64), (130, 16, 16, 64), (72, 32, 32, 128), (100, 16, 16, 256), (300, 8, 8, 512), (37, 8, 8, 256)]:
    for mode in ("plain", "transform", "residual", "masked"):
        g = torch.Generator(device=dev); g.manual_seed(N * 131 + H)
        x = torch.randn(N, H, W, Cc, device=dev, generator=g)
        w = torch.randn(Cc, 3, 3, Cc, device=dev, generator=g) * 0.05
        tr = res = mask = None
        if mode == "transform":
            tr = [torch.randn(Cc, device=dev, generator=g), torch.rand(Cc, device=dev, generator=g) + 0.5,
                  torch.randn(Cc, device=dev, generator=g), torch.randn(Cc, device=dev, generator=g)]
        if mode in ("residual", "masked"):
            res = torch.randn(N, H, W, Cc, device=dev, generator=g)
        if mode == "masked":
            mask = torch.randint(0, 16, (N * H * W * Cc // 4,), device=dev, dtype=torch.uint8, generator=g)
        a = run(libs[0], x, w, N, H, W, Cc, tr, res, mask)
        b = run(libs[1], x, w, N, H, W, Cc, tr, res, mask)
        names = ("out", "emit", "mean", "invstd", "running_mean", "running_var")
        diffs = [n for n, u, v in zip(names, a, b) if u is not None and not torch.equal(u, v)]
        print(f"{N}x{H}x{W}x{Cc} {mode}: {'identical' if not diffs else 'DIFFERENT: ' + ', '.join(diffs)}")
        bad += bool(diffs)
sys.exit(1 if bad else 0)
