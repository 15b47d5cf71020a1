"""Tensor-level wrappers over the C ABI: allocate outputs with torch, launch on the current stream."""
import os

import torch

from . import _hip


PROFILE = None      # bench.py sets this to a list: every launch is then bracketed by HIP events on its stream


def _new(shape, like):
    return torch.empty(shape, device=like.device, dtype=torch.float32)


def _newh(shape, like):
    """A half tensor: the activations of the precision-16 step are stored as torch.autocast stores them."""
    return torch.empty(shape, device=like.device, dtype=torch.float16)


def _is_h(t):
    return t is not None and t.dtype == torch.float16


def _p(t, allow_none=False):
    """Device pointer of a contiguous fp32 OR half tensor (the `_h` entry points take void*)."""
    return _hip.ptr(t, allow_none, torch.float16 if _is_h(t) else torch.float32)


def _run(kernel, flops, nbytes, rc_fn, exec_flops=None, tile=None):
    """Launch through the C ABI; when PROFILE is on, bracket the launch with events on the launch stream.
    flops = ALGORITHMIC FLOPs of the op; exec_flops = the FLOPs the kernel really issues to the matrix cores when that
    is fewer (position-major convs skip the taps that fall into the zero padding); tile = a callable returning the
    instantiation's tile code (ssad_conv_igemm_tile), evaluated only while profiling."""
    if PROFILE is None:
        _hip.check(rc_fn())
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _hip.check(rc_fn())
    e1.record()
    PROFILE.append((kernel, flops, nbytes, e0, e1, flops if exec_flops is None else exec_flops, tile() if tile else None))


# Launches nobody on the critical path of a training step waits for -- the slab reduction behind a weight-gradient kernel, the
# weight gradients of the head's small linear layers, global-average-pool rows that only the head reads -- may leave the main
# stream: TrainEngine sets ASIDE to an object whose launch(fn, keep) runs fn on a second HIP stream that has just been made to
# wait for the main one, and keeps `keep` (the operands) alive until the streams are joined.  Recorded into a hipGraph this is
# a fork / join: the small kernel becomes a parallel branch beside the next big kernel instead of a ~5-12 us link in the chain.
ASIDE = None


def _run_aside(kernel, flops, nbytes, rc_fn, keep=()):
    if ASIDE is None or PROFILE is not None:
        return _run(kernel, flops, nbytes, rc_fn)
    ASIDE.launch(lambda: _hip.check(rc_fn()), keep)


# Slab reductions of the weight-gradient kernels, collected while PENDING_REDUCE is a list (TrainEngine sets it for a training step)
# and run together by flush_reductions() where the gradients are first needed: one launch (ssad_wgrad_reduce_batch) instead of one
# per layer.  Entries keep their slabs alive until the flush.
PENDING_REDUCE = None


def _wgrad_reduce(slab, dw_out, splits, cout, kpad, kh, kw, cin, to_oihw, accumulate):
    kreal = kh * kw * cin
    if (PENDING_REDUCE is not None and PROFILE is None and not to_oihw and not accumulate and kreal % 4 == 0 and kpad % 4 == 0
            and slab.data_ptr() % 16 == 0 and dw_out.data_ptr() % 16 == 0):
        PENDING_REDUCE.append((slab, dw_out, int(splits), int(cout), int(kpad), int(kreal)))
        return
    _run_aside("wgrad_reduce", 0.0, 4.0 * slab.numel(),
               lambda: _hip.lib().ssad_wgrad_reduce(_hip.ptr(slab), _hip.ptr(dw_out), splits, cout, kpad, kh, kw, cin, int(to_oihw),
                                                    int(accumulate), _hip.stream()), keep=(slab,))


def flush_reductions():
    """Run the collected slab reductions (one launch per 24) and release their slabs."""
    if not PENDING_REDUCE:
        return
    import ctypes
    desc = []
    for slab, out, splits, cout, kpad, kreal in PENDING_REDUCE:
        desc += [slab.data_ptr(), out.data_ptr(), splits, cout, kpad, kreal]
    arr = (ctypes.c_int64 * len(desc))(*desc)
    _hip.check(_hip.lib().ssad_wgrad_reduce_batch(arr, len(PENDING_REDUCE), _hip.stream()))
    del PENDING_REDUCE[:]


def drain_profile():
    """-> [{kernel, flops, exec_flops, bytes, ms}] for every launch recorded since PROFILE was set."""
    global PROFILE
    recs = PROFILE or []
    torch.cuda.synchronize()
    out = [{"kernel": k, "flops": f, "bytes": b, "ms": e0.elapsed_time(e1), "exec_flops": x, "tile": t}
           for k, f, b, e0, e1, x, t in recs]
    PROFILE = [] if PROFILE is not None else None
    return out


def _inbounds_taps(h, w, kh, kw, stride, pad):
    """Sum over output positions of the filter taps that land inside the image (what a position-major conv executes)."""
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    ny = sum(sum(1 for ky in range(kh) if 0 <= oy * stride - pad + ky < h) for oy in range(ho))
    nx = sum(sum(1 for kx in range(kw) if 0 <= ox * stride - pad + kx < w) for ox in range(wo))
    return ny * nx


def igemm_tile_name(n, h, w, cin, cout, kh, kw, stride, pad, hwnc):
    """Template arguments of the exact-fp32 implicit-GEMM instantiation a problem runs on (csrc/conv_igemm.hip: BM, BN, TM, TN, BK)."""
    if not hwnc and kh == kw == h == w == 1 and n <= _hip.lib().ssad_linear_small_max_rows():
        return "<linear_small: 32 x 32 tiles, csrc/linear_small.hip>"
    code = _hip.lib().ssad_conv_igemm_tile(n, h, w, cin, cout, kh, kw, stride, pad, int(hwnc))
    pos, code = code < 0, abs(code)
    bm, bn, bk = code // 100000, code // 100 % 1000, code % 100
    tm, tn = {(256, 64): (2, 2), (128, 64): (1, 2), (256, 128): (2, 2), (128, 256): (2, 4), (64, 64): (1, 1)}.get((bm, bn), (2, 2))
    return f"<{bm},{bn},{tm},{tn},{bk},1,{'true' if pos else 'false'}>"


def _kname(base, mode):
    """Profile label of an MFMA kernel by operand mode (False/0 fp32, True/1 bf16, 2 fp16, 3 / 6 split-bf16)."""
    return base + {0: "_f32", 1: "_bf16", 2: "_f16", 3: "_x3", 6: "_x6"}[int(mode)]


def repack_oihw_to_ohwi(w):
    o, i, kh, kw = w.shape
    out = _new((o, kh, kw, i), w)
    _hip.check(_hip.lib().ssad_repack_oihw_to_ohwi(_hip.ptr(w), _hip.ptr(out), o, i, kh, kw, _hip.stream()))
    return out


def repack_ohwi_to_oihw(w):
    o, kh, kw, i = w.shape
    out = _new((o, i, kh, kw), w)
    _hip.check(_hip.lib().ssad_repack_ohwi_to_oihw(_hip.ptr(w), _hip.ptr(out), o, i, kh, kw, _hip.stream()))
    return out


def pack_stem_weight(w):
    """conv1's filter in the stem kernel's K order; w OIHW [64][3][7][7], or OHWI [64][7][7][3] (the parameter arena's layout)."""
    out = _new((168, 64), w)
    if tuple(w.shape) == (64, 7, 7, 3):
        _hip.check(_hip.lib().ssad_pack_stem_weight_ohwi(_hip.ptr(w), _hip.ptr(out), _hip.stream()))
        return out
    assert tuple(w.shape) == (64, 3, 7, 7)
    _hip.check(_hip.lib().ssad_pack_stem_weight(_hip.ptr(w), _hip.ptr(out), _hip.stream()))
    return out


def stem_geometry(H, W, patch_dim, patch_stride):
    """(samples per image, Hv, Wv, Ho, Wo) -- window + nearest-resize rule of models.py:211-219."""
    if patch_dim:
        p = ((H - patch_dim) // patch_stride + 1) * ((W - patch_dim) // patch_stride + 1)
        wh = ww = patch_dim
    else:
        p, wh, ww = 1, H, W
    hv, wv = (64, 64) if (wh < 64 or ww < 64) else (wh, ww)
    return p, hv, wv, (hv - 1) // 2 + 1, (wv - 1) // 2 + 1


def stem_fwd(img, wk, scale, shift, relu=True, patch_dim=0, patch_stride=0, hwnc=False, resize_to=None):
    """-> [N][Ho][Wo][64], or the position-major [Ho][Wo][N][64] when hwnc.
    resize_to = (Hv, Wv): the whole image nearest-resized to Hv x Wv inside the loader (floor(v * H / Hv): F.interpolate's 'nearest')
    instead of the reference's to-64 rule -- the per-image dense map of the patch-scoring pass takes the 2x upsample this way."""
    b, c, h, w = img.shape
    assert c == 3
    p, hv, wv, ho, wo = stem_geometry(h, w, patch_dim, patch_stride)
    if resize_to is not None:
        assert not patch_dim
        hv, wv = resize_to
        ho, wo = (hv - 1) // 2 + 1, (wv - 1) // 2 + 1
    n = b * p
    out = _new((ho, wo, n, 64) if hwnc else (n, ho, wo, 64), img)
    _run("stem_conv7x7", 2.0 * n * ho * wo * 64 * 147, 4.0 * (b * 3 * h * w + n * ho * wo * 64),
         lambda: _hip.lib().ssad_stem_fwd(_hip.ptr(img), b, h, w, patch_dim, patch_stride, hv, wv, _hip.ptr(wk),
                                          _hip.ptr(scale, True), _hip.ptr(shift, True), int(relu), int(hwnc),
                                          _hip.ptr(out), _hip.stream()))
    return out


def stem_fwd_stats(img, wk, eps, momentum, running_mean, running_var):
    """Training stem conv: -> (z [B][Ho][Wo][64], mean, invstd) with the BatchNorm statistics taken from the accumulators."""
    b, c, h, w = img.shape
    assert c == 3
    _, hv, wv, ho, wo = stem_geometry(h, w, 0, 0)
    out = _new((b, ho, wo, 64), img)
    mean, invstd = _new((64,), img), _new((64,), img)
    ws = torch.empty(_hip.lib().ssad_stem_stats_rows() * 128, device=img.device, dtype=torch.float64)
    _run("stem_conv7x7", 2.0 * b * ho * wo * 64 * 147, 4.0 * (b * 3 * h * w + b * ho * wo * 64),
         lambda: _hip.lib().ssad_stem_fwd_stats(_hip.ptr(img), b, h, w, hv, wv, _hip.ptr(wk), _hip.ptr(out), eps, momentum,
                                                _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(running_mean, True),
                                                _hip.ptr(running_var, True), ws.data_ptr(), _hip.stream()))
    return out, mean, invstd


def stem_fwd_stats16(img, w_oihw, eps, momentum, running_mean, running_var, mode, out_half=False):
    """The training stem conv with fp16 (mode 2) or bf16 (mode 1) operands and fp32 accumulation (csrc/stem16.hip):
    -> (z [B][Ho][Wo][64] fp32, mean, invstd), statistics from the accumulators as in stem_fwd_stats.
    out_half (mode 2): z is stored as halves and the statistics are those of the stored halves."""
    b, c, h, w = img.shape
    ohwi = tuple(w_oihw.shape) == (64, 7, 7, 3)           # the parameter arena's layout: packed without an OIHW copy
    assert c == 3 and (ohwi or tuple(w_oihw.shape) == (64, 3, 7, 7)) and int(mode) in (1, 2)
    assert h >= 64 and w >= 64, "images below 64 x 64 are resized first (models.py:217-219): that is the fp32 stem's loader"
    lib = _hip.lib()
    wk = torch.empty(14 * 64 * 16, device=img.device, dtype=torch.float16 if int(mode) == 2 else torch.bfloat16)
    _hip.check((lib.ssad_pack_stem_weight16_ohwi if ohwi else lib.ssad_pack_stem_weight16)(_hip.ptr(w_oihw), wk.data_ptr(), int(int(mode) == 2),
                                                                                             _hip.stream()))
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    mean, invstd = _new((64,), img), _new((64,), img)
    ws = torch.empty(lib.ssad_stem_stats_rows() * 128, device=img.device, dtype=torch.float64)
    if out_half:
        assert int(mode) == 2
        out = _newh((b, ho, wo, 64), img)
        _run("stem_conv7x7_h16", 2.0 * b * ho * wo * 64 * 147, 4.0 * b * 3 * h * w + 2.0 * b * ho * wo * 64,
             lambda: lib.ssad_stem_fwd_stats16_h(_hip.ptr(img), b, h, w, wk.data_ptr(), out.data_ptr(), eps, momentum, _hip.ptr(mean),
                                                 _hip.ptr(invstd), _hip.ptr(running_mean, True), _hip.ptr(running_var, True),
                                                 ws.data_ptr(), _hip.stream()))
        return out, mean, invstd
    out = _new((b, ho, wo, 64), img)
    _run(_kname("stem_conv7x7", mode), 2.0 * b * ho * wo * 64 * 147, 4.0 * (b * 3 * h * w + b * ho * wo * 64),
         lambda: lib.ssad_stem_fwd_stats16(_hip.ptr(img), b, h, w, wk.data_ptr(), _hip.ptr(out), eps, momentum, _hip.ptr(mean),
                                           _hip.ptr(invstd), _hip.ptr(running_mean, True), _hip.ptr(running_var, True),
                                           ws.data_ptr(), int(int(mode) == 2), _hip.stream()))
    return out, mean, invstd


def pack_stem_weight_folded(w):
    assert tuple(w.shape) == (64, 3, 7, 7)
    out = _new((24, 2, 64), w)
    _hip.check(_hip.lib().ssad_pack_stem_weight_folded(_hip.ptr(w), _hip.ptr(out), _hip.stream()))
    return out


def stem_patch_pool_fwd(img, wf, scale, shift, patch_stride=8, hwnc=False, skip=None):
    """32x32 windows (stride patch_stride) of img [B][3][H][W] -> pooled stem output [N][16][16][64] / [16][16][N][64]:
    window + 2x nearest upsample + conv7x7/2 + affine + ReLU + max-pool in one kernel.  skip = (lo, hi): the pooled positions
    lo <= py, px <= hi of every patch are left unwritten (nobody reads them when layer1 is shared between overlapping patches)."""
    b, c, h, w = img.shape
    n = b * ((h - 32) // patch_stride + 1) * ((w - 32) // patch_stride + 1)
    out = _new((16, 16, n, 64) if hwnc else (n, 16, 16, 64), img)
    lo, hi = skip if skip is not None else (1, 0)
    _run("stem_patch_pool", 2.0 * n * 32 * 32 * 64 * 147, 4.0 * (img.numel() + out.numel()),
         lambda: _hip.lib().ssad_stem_patch_pool_fwd_ring(_hip.ptr(img), b, h, w, patch_stride, _hip.ptr(wf), _hip.ptr(scale, True),
                                                          _hip.ptr(shift, True), int(hwnc), lo, hi, _hip.ptr(out), _hip.stream()))
    return out


def stem_patch_border_fwd(img, wf, scale, shift, patch_stride=8):
    """Pooled rows / columns 0, 1 and 15 of every 32x32 window's stem map (the positions its own zero padding reaches), written into a
    position-major [16][16][N][64] buffer whose other positions stay unwritten: see patch_gather_hwnc for rows / columns 2-3, 13-14."""
    b, c, h, w = img.shape
    n = b * ((h - 32) // patch_stride + 1) * ((w - 32) // patch_stride + 1)
    out = _new((16, 16, n, 64), img)
    _run("stem_patch_pool", 2.0 * n * 13 * 32 * 64 * 147, 4.0 * (img.numel() + out.numel() * 87 // 256),
         lambda: _hip.lib().ssad_stem_patch_border_fwd(_hip.ptr(img), b, h, w, patch_stride, _hip.ptr(wf), _hip.ptr(scale, True),
                                                       _hip.ptr(shift, True), _hip.ptr(out), _hip.stream()))
    return out


def maxpool3x3s2_fwd(x, hwnc=False):
    if hwnc:
        h, w, n, c = x.shape
        out = _new(((h - 1) // 2 + 1, (w - 1) // 2 + 1, n, c), x)
    else:
        n, h, w, c = x.shape
        out = _new((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), x)
    _run("maxpool3x3s2", 0.0, 4.0 * (x.numel() + out.numel()),
         lambda: _hip.lib().ssad_maxpool3x3s2_fwd(_hip.ptr(x), _hip.ptr(out), n, h, w, c, int(hwnc), _hip.stream()))
    return out


def _split_fwd(mode):
    return _hip.lib().ssad_conv_igemm_fwd_x6 if mode == 6 else _hip.lib().ssad_conv_igemm_fwd_x3


def conv_fwd_hwnc(x, w_ohwi, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0, x3=False):
    """Position-major activations: x [H][W][N][Cin] -> [Ho][Wo][N][Cout] (patch-scoring trunk layout).
    x3: split-bf16 products (SSAD_MATH=bf16x3)."""
    h, w, n, cin = x.shape
    cout, kh, kw, cin2 = w_ohwi.shape
    assert cin == cin2
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    out = _new((ho, wo, n, cout), x)
    nb = 4.0 * (x.numel() + out.numel() * (2 if residual is not None else 1) + w_ohwi.numel())
    if x3:                                       # True / 3: bf16x3, 6: bf16x6
        _run("conv_igemm_x6" if x3 == 6 else "conv_igemm_x3", 2.0 * out.numel() * kh * kw * cin, nb,
             lambda: _split_fwd(x3)(_hip.ptr(x), _hip.ptr(w_ohwi), _hip.ptr(out), _hip.ptr(scale, True),
                                    _hip.ptr(shift, True), _hip.ptr(residual, True), int(relu), n, h, w,
                                    cin, cout, kh, kw, stride, pad, 1, _hip.stream()))
        return out
    _run("conv_igemm_pos_f32", 2.0 * out.numel() * kh * kw * cin, nb,          # the position-major instantiations (POS = true)
         lambda: _hip.lib().ssad_conv_igemm_fwd_hwnc(_hip.ptr(x), _hip.ptr(w_ohwi), _hip.ptr(out), _hip.ptr(scale, True),
                                                     _hip.ptr(shift, True), _hip.ptr(residual, True), int(relu), n, h, w,
                                                     cin, cout, kh, kw, stride, pad, _hip.stream()),
         exec_flops=2.0 * n * cout * cin * _inbounds_taps(h, w, kh, kw, stride, pad) if PROFILE is not None else None,
         tile=lambda: igemm_tile_name(n, h, w, cin, cout, kh, kw, stride, pad, 1))
    return out


def conv_fwd_hwnc_ring(x, w_ohwi, scale, shift, residual, relu, skip_lo, skip_hi):
    """3x3 / stride 1 / pad 1 conv over position-major activations, ONLY the output positions outside the square
    skip_lo <= oy, ox <= skip_hi (the others are left for patch_gather_hwnc).  x [H][W][N][Cin] -> [H][W][N][Cout]."""
    h, w, n, cin = x.shape
    cout, kh, kw, cin2 = w_ohwi.shape
    assert cin == cin2 and (kh, kw) == (3, 3) and 0 <= skip_lo <= skip_hi < min(h, w)
    out = _new((h, w, n, cout), x)
    side = skip_hi - skip_lo + 1
    ring = h * w - side * side
    taps = _inbounds_taps(h, w, 3, 3, 1, 1) - 9 * side * side if PROFILE is not None else 0       # the skipped square has all nine taps
    _run("conv_igemm_pos_f32", 2.0 * n * ring * cout * 9 * cin, 4.0 * (x.numel() + n * ring * cout * (2 if residual is not None else 1) + w_ohwi.numel()),
         lambda: _hip.lib().ssad_conv_igemm_fwd_hwnc_ring(_hip.ptr(x), _hip.ptr(w_ohwi), _hip.ptr(out), _hip.ptr(scale, True),
                                                          _hip.ptr(shift, True), _hip.ptr(residual, True), int(relu), n, h, w,
                                                          cin, cout, 3, 3, 1, 1, skip_lo, skip_hi, _hip.stream()),
         exec_flops=2.0 * n * cout * cin * taps if PROFILE is not None else None,
         tile=lambda: igemm_tile_name(n, h, w, cin, cout, 3, 3, 1, 1, 2))
    return out


def patch_gather_hwnc(dense, out, prow, pcol, shift, lo, hi, ilo=1, ihi=0):
    """out[u][v][n][:] = dense[b][shift * pr + u][shift * pc + v][:] for lo <= u, v <= hi (n = (b * prow + pr) * pcol + pc): the
    positions of every patch's map that equal the per-image dense map.  dense NHWC [B][Hd][Wd][C], out [H][W][N][C], in place.
    ilo <= ihi: the inner square ilo <= u, v <= ihi is left untouched (nobody reads it)."""
    b, hd, wd, c = dense.shape
    h, w, n, c2 = out.shape
    assert c == c2 and n == b * prow * pcol
    side, iside = hi - lo + 1, max(ihi - ilo + 1, 0)
    _run("patch_gather", 0.0, 8.0 * n * (side * side - iside * iside) * c,
         lambda: _hip.lib().ssad_patch_gather_hwnc_band(_hip.ptr(dense), _hip.ptr(out), b, prow, pcol, shift, hd, wd, c, h, w, lo, hi,
                                                        ilo, ihi, _hip.stream()))
    return out


def conv_fwd(x, w_ohwi, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0, bf16=False):
    """x NHWC [N][H][W][Cin]; w OHWI [Cout][KH][KW][Cin] -> NHWC [N][Ho][Wo][Cout].
    bf16=True: operands rounded to bf16 in the loader (fp32 storage / accumulate), the Trainer(precision=16) path;
    bf16=3: split-bf16 products (hi*hi + hi*lo + lo*hi), fp32-class accuracy from the bf16 matrix cores."""
    n, h, w, cin = x.shape
    cout, kh, kw, cin2 = w_ohwi.shape
    assert cin == cin2
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    out = _new((n, ho, wo, cout), x)
    nb = 4.0 * (x.numel() + out.numel() * (2 if residual is not None else 1) + w_ohwi.numel())
    if bf16 in (3, 6):
        _run("conv_igemm_x6" if bf16 == 6 else "conv_igemm_x3", 2.0 * out.numel() * kh * kw * cin, nb,
             lambda: _split_fwd(bf16)(_hip.ptr(x), _hip.ptr(w_ohwi), _hip.ptr(out), _hip.ptr(scale, True),
                                      _hip.ptr(shift, True), _hip.ptr(residual, True), int(relu), n, h, w,
                                      cin, cout, kh, kw, stride, pad, 0, _hip.stream()))
        return out
    fn = (_hip.lib().ssad_conv_igemm_fwd_f16 if bf16 == 2 else _hip.lib().ssad_conv_igemm_fwd_bf16 if bf16 else
          _hip.lib().ssad_conv_igemm_fwd)
    _run(_kname("conv_igemm", bf16), 2.0 * out.numel() * kh * kw * cin, nb,
         lambda: fn(_hip.ptr(x), _hip.ptr(w_ohwi), _hip.ptr(out), _hip.ptr(scale, True), _hip.ptr(shift, True),
                    _hip.ptr(residual, True), int(relu), n, h, w, cin, cout, kh, kw, stride, pad, _hip.stream()),
         tile=None if bf16 else (lambda: igemm_tile_name(n, h, w, cin, cout, kh, kw, stride, pad, 0)))
    return out


def conv_fwd_stats(x, w_ohwi, eps, momentum, running_mean, running_var, stride=1, pad=0, bf16=False):
    """Bias-free conv + the train-mode BatchNorm statistics of its output in one pass.  Returns (z, mean, invstd)."""
    n, h, w, cin = x.shape
    cout, kh, kw, cin2 = w_ohwi.shape
    assert cin == cin2
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    mean, invstd = _new((cout,), x), _new((cout,), x)
    lib = _hip.lib()
    ws = torch.empty(lib.ssad_conv_stats_workspace(n, ho, wo, cout), device=x.device, dtype=torch.float64)
    if _is_h(x):                 # precision-16 step with half tensors: x, the weights and z are halves
        assert _is_h(w_ohwi) and bf16 == 2
        out = _newh((n, ho, wo, cout), x)
        _run("conv_igemm_h16", 2.0 * out.numel() * kh * kw * cin, 2.0 * (x.numel() + out.numel() + w_ohwi.numel()),
             lambda: lib.ssad_conv_igemm_fwd_stats_h(x.data_ptr(), w_ohwi.data_ptr(), out.data_ptr(), n, h, w, cin, cout, kh, kw,
                                                     stride, pad, eps, momentum, _hip.ptr(mean), _hip.ptr(invstd),
                                                     _hip.ptr(running_mean, True), _hip.ptr(running_var, True), ws.data_ptr(),
                                                     _hip.stream()))
        return out, mean, invstd
    out = _new((n, ho, wo, cout), x)
    nb = 4.0 * (x.numel() + out.numel() + w_ohwi.numel())
    _run(_kname("conv_igemm", bf16), 2.0 * out.numel() * kh * kw * cin, nb,
         lambda: lib.ssad_conv_igemm_fwd_stats(_hip.ptr(x), _hip.ptr(w_ohwi), _hip.ptr(out), n, h, w, cin, cout, kh, kw,
                                               stride, pad, int(bf16), eps, momentum, _hip.ptr(mean), _hip.ptr(invstd),
                                               _hip.ptr(running_mean, True), _hip.ptr(running_var, True), ws.data_ptr(),
                                               _hip.stream()),
         tile=None if bf16 else (lambda: igemm_tile_name(n, h, w, cin, cout, kh, kw, stride, pad, 0)))
    return out, mean, invstd


def conv3x3_c64(x, w_ohwi, residual=None, transform=None, emit=False, stats=None, res_mask=None, bf16=0):
    """Halo-tile 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels (layer1 forward and input-gradient convs).
    x NHWC [N][H][W][64], w OHWI [64][3][3][64].  transform = (mean, invstd, gamma, beta): the input is taken through
    relu(bn(x)) while it is staged (emit=True also returns that activation).  stats = (eps, momentum, running_mean,
    running_var): also the train-mode BatchNorm statistics of the output.  Returns out[, emitted][, mean, invstd]."""
    n, h, w, c = x.shape
    assert c == 64 and tuple(w_ohwi.shape) == (64, 3, 3, 64)
    lib = _hip.lib()
    half = _is_h(x)
    assert residual is None or tuple(residual.shape) == (n, h, w, 64), "residual must have the output's shape"
    out = _newh((n, h, w, 64), x) if half else _new((n, h, w, 64), x)
    em = torch.empty_like(x) if emit else None
    tr = transform if transform is not None else (None, None, None, None)
    mean = invstd = ws = None
    eps = mom = 0.0
    rm = rv = None
    if stats is not None:
        eps, mom, rm, rv = stats
        mean, invstd = _new((64,), x), _new((64,), x)
        ws = torch.empty(lib.ssad_conv3x3_c64_stats_rows(n, h, w) * 128, device=x.device, dtype=torch.float64)
    nb = 4.0 * (x.numel() + out.numel() * (2 if residual is not None else 1) + (out.numel() if emit else 0) + w_ohwi.numel())
    op = int(bf16)
    assert op in (0, 1, 2), "conv3x3_c64: exact fp32, or bf16 (1) / fp16 (2) operands"
    if half:
        assert op == 2 and res_mask is None and (residual is None or _is_h(residual))
        _run("conv_c64_h16", 2.0 * out.numel() * 9 * 64, nb / 2 + 2.0 * w_ohwi.numel(),
             lambda: lib.ssad_conv3x3_c64_h(x.data_ptr(), _hip.ptr(w_ohwi), out.data_ptr(), _p(residual, True), _hip.ptr(tr[0], True),
                                            _hip.ptr(tr[1], True), _hip.ptr(tr[2], True), _hip.ptr(tr[3], True), _p(em, True),
                                            n, h, w, ws.data_ptr() if ws is not None else None, eps, mom, _hip.ptr(mean, True),
                                            _hip.ptr(invstd, True), _hip.ptr(rm, True), _hip.ptr(rv, True), _hip.stream()))
    else:
        _run(_kname("conv_c64", op), 2.0 * out.numel() * 9 * 64, nb,
             lambda: lib.ssad_conv3x3_c64_op(_hip.ptr(x), _hip.ptr(w_ohwi), _hip.ptr(out), _hip.ptr(residual, True),
                                             res_mask.data_ptr() if res_mask is not None else None, _hip.ptr(tr[0], True),
                                             _hip.ptr(tr[1], True), _hip.ptr(tr[2], True), _hip.ptr(tr[3], True), _hip.ptr(em, True),
                                             n, h, w, ws.data_ptr() if ws is not None else None, eps, mom, _hip.ptr(mean, True),
                                             _hip.ptr(invstd, True), _hip.ptr(rm, True), _hip.ptr(rv, True), op, _hip.stream()))
    res = [out]
    if emit:
        res.append(em)
    if stats is not None:
        res += [mean, invstd]
    return res[0] if len(res) == 1 else tuple(res)


def conv3x3_h(x, w_ohwi, residual=None, transform=None, emit=False, stats=None):
    """3x3 / stride 1 / pad 1 convolution over half tensors (csrc/conv16.hip): x NHWC [N][H][W][Cin] halves, w OHWI halves
    [Cout][3][3][Cin] -> [N][H][W][Cout] halves (+ residual).  transform = (mean, invstd, gamma, beta) of the PRODUCING layer: x is
    taken through relu(bn(x)) while it is staged (emit=True also returns that activation, for the weight gradient).  stats = (eps,
    momentum, running_mean, running_var): also the train-mode BatchNorm statistics of the stored output.
    Returns out[, emitted][, mean, invstd]."""
    n, h, w, cin = x.shape
    cout = w_ohwi.shape[0]
    assert _is_h(x) and _is_h(w_ohwi) and tuple(w_ohwi.shape) == (cout, 3, 3, cin)
    assert residual is None or (_is_h(residual) and tuple(residual.shape) == (n, h, w, cout)), "residual must have the output's shape"
    lib = _hip.lib()
    out = _newh((n, h, w, cout), x)
    em = torch.empty_like(x) if emit else None
    tr = transform if transform is not None else (None, None, None, None)
    mean = invstd = ws = None
    eps = mom = 0.0
    rm = rv = None
    if stats is not None:
        eps, mom, rm, rv = stats
        mean, invstd = _new((cout,), x), _new((cout,), x)
        ws = torch.empty(lib.ssad_conv3x3_h_stats_rows(n, h, w, cout) * 2 * cout, device=x.device, dtype=torch.float64)
    nb = 2.0 * (x.numel() + out.numel() * (2 if residual is not None else 1) + (x.numel() if emit else 0) + w_ohwi.numel())
    _run("conv3x3_h16", 2.0 * out.numel() * 9 * cin, nb,
         lambda: lib.ssad_conv3x3_h(x.data_ptr(), w_ohwi.data_ptr(), out.data_ptr(), _p(residual, True), _hip.ptr(tr[0], True),
                                    _hip.ptr(tr[1], True), _hip.ptr(tr[2], True), _hip.ptr(tr[3], True), _p(em, True), n, h, w, cin, cout,
                                    ws.data_ptr() if ws is not None else None, eps, mom, _hip.ptr(mean, True), _hip.ptr(invstd, True),
                                    _hip.ptr(rm, True), _hip.ptr(rv, True), _hip.stream()))
    res = [out]
    if emit:
        res.append(em)
    if stats is not None:
        res += [mean, invstd]
    return res[0] if len(res) == 1 else tuple(res)


def conv3x3_hw_ok(n, h, w, cin, cout, f32=False):
    """Whether ssad_conv3x3_hw / _fw (csrc/conv16w.hip: register-fed filters, 128 x 64 wave tiles) takes a launch of this shape."""
    lib = _hip.lib()
    return bool((lib.ssad_conv3x3_fw_ok if f32 else lib.ssad_conv3x3_hw_ok)(n, h, w, cin, cout))


def conv3x3_hw_pack(src_f32, entries, out=None, f32=False):
    """Packs 3x3 filters for conv3x3_hw in ONE launch.  src_f32: a flat fp32 tensor holding OHWI filters; entries: iterable of
    (source offset in floats, Cout, Cin of the conv that will run on the packed filter, flip) -- flip: the source is the OHWI filter
    [Cin][3][3][Cout] of the forward conv whose input gradient this is.  f32: the pack of the exact-fp32 form (floats).
    Returns (flat tensor, [offset of each packed filter])."""
    import ctypes
    desc, offs, off = [], [], 0
    for (so, o, i, flip) in entries:
        desc += [int(so), off, int(o), int(i), int(bool(flip))]
        offs.append(off)
        off += o * 9 * i
    dt = torch.float32 if f32 else torch.float16
    if out is None:
        out = torch.empty(off, device=src_f32.device, dtype=dt)
    assert out.numel() >= off and out.dtype == dt and src_f32.dtype == torch.float32 and src_f32.is_contiguous()
    arr = (ctypes.c_int64 * len(desc))(*desc)
    lib = _hip.lib()
    _hip.check((lib.ssad_conv3x3_fw_pack_batch if f32 else lib.ssad_conv3x3_hw_pack_batch)(_hip.ptr(src_f32), out.data_ptr(), arr, len(offs),
                                                                                            _hip.stream()))
    return out, offs


def conv3x3_hw(x, w_packed, cout, residual=None, transform=None, emit=False, stats=None, res_mask=None):
    """conv3x3_h with the filter packed by conv3x3_hw_pack (a flat tensor of cout * 9 * cin values); half tensors, or -- x fp32 -- the
    exact-fp32 instantiation (res_mask: the residual is gated by the nibble mask of bn_apply_fwd_mask).  The launch must satisfy
    conv3x3_hw_ok.  Returns out[, emitted][, mean, invstd]."""
    n, h, w, cin = x.shape
    f32 = x.dtype == torch.float32
    assert (f32 or _is_h(x)) and w_packed.dtype == x.dtype and w_packed.numel() == cout * 9 * cin and w_packed.is_contiguous()
    assert x.is_contiguous() and conv3x3_hw_ok(n, h, w, cin, cout, f32), "shape outside ssad_conv3x3_hw_ok"
    assert residual is None or (residual.dtype == x.dtype and tuple(residual.shape) == (n, h, w, cout) and residual.is_contiguous()), \
        "residual must have the output's shape"
    assert res_mask is None or (residual is not None and res_mask.numel() == n * h * w * cout // 4 and res_mask.dtype == torch.uint8)
    lib = _hip.lib()
    out = torch.empty((n, h, w, cout), device=x.device, dtype=x.dtype)
    em = torch.empty_like(x) if emit else None
    tr = transform if transform is not None else (None, None, None, None)
    mean = invstd = ws = None
    eps = mom = 0.0
    rm = rv = None
    if stats is not None:
        eps, mom, rm, rv = stats
        mean, invstd = _new((cout,), x), _new((cout,), x)
        ws = torch.empty(lib.ssad_conv3x3_hw_stats_rows(n, h, w, cout) * 2 * cout, device=x.device, dtype=torch.float64)
    nb = x.element_size() * (x.numel() + out.numel() * (2 if residual is not None else 1) + (x.numel() if emit else 0) + w_packed.numel())
    wsp = ws.data_ptr() if ws is not None else None
    if f32:
        _run("conv3x3_fw32", 2.0 * out.numel() * 9 * cin, float(nb),
             lambda: lib.ssad_conv3x3_fw(x.data_ptr(), w_packed.data_ptr(), out.data_ptr(), _p(residual, True),
                                         res_mask.data_ptr() if res_mask is not None else None, _hip.ptr(tr[0], True), _hip.ptr(tr[1], True),
                                         _hip.ptr(tr[2], True), _hip.ptr(tr[3], True), _p(em, True), n, h, w, cin, cout, wsp, eps, mom,
                                         _hip.ptr(mean, True), _hip.ptr(invstd, True), _hip.ptr(rm, True), _hip.ptr(rv, True), _hip.stream()))
    else:
        _run("conv3x3_hw16", 2.0 * out.numel() * 9 * cin, float(nb),
             lambda: lib.ssad_conv3x3_hw(x.data_ptr(), w_packed.data_ptr(), out.data_ptr(), _p(residual, True),
                                         res_mask.data_ptr() if res_mask is not None else None, _hip.ptr(tr[0], True),
                                         _hip.ptr(tr[1], True), _hip.ptr(tr[2], True), _hip.ptr(tr[3], True), _p(em, True), n, h, w, cin, cout,
                                         wsp, eps, mom, _hip.ptr(mean, True), _hip.ptr(invstd, True), _hip.ptr(rm, True), _hip.ptr(rv, True),
                                         _hip.stream()))
    res = [out]
    if emit:
        res.append(em)
    if stats is not None:
        res += [mean, invstd]
    return res[0] if len(res) == 1 else tuple(res)


def conv3x3_fw_eval_ok(n, h, w, cin, cout):
    return bool(_hip.lib().ssad_conv3x3_fw_eval_ok(n, h, w, cin, cout))


def conv3x3_fw_pack_scaled(w_ohwi, scale):
    """The filter of an inference conv in the register-fed kernel's fragment order, the folded BatchNorm's scale multiplied in."""
    cout, kh, kw, cin = w_ohwi.shape
    assert (kh, kw) == (3, 3) and w_ohwi.is_contiguous() and (scale is None or scale.numel() == cout)
    out = torch.empty(cout * 9 * cin, device=w_ohwi.device, dtype=torch.float32)
    _hip.check(_hip.lib().ssad_conv3x3_fw_pack_scaled(_hip.ptr(w_ohwi), _hip.ptr(scale, True), _hip.ptr(out), cout, cin, _hip.stream()))
    return out


def conv3x3_fw_eval(x, w_packed, cout, shift, residual=None, relu=False, out_hwnc=False):
    """Inference 3x3 / stride 1 conv on the register-fed kernel (csrc/conv16w.hip, T = float): act(conv(x) + shift (+ residual)), x and
    residual NHWC, the output NHWC or position-major [H][W][N][C]; the filter from conv3x3_fw_pack_scaled."""
    n, h, w, cin = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and w_packed.numel() == cout * 9 * cin and shift.numel() == cout
    assert residual is None or (tuple(residual.shape) == (n, h, w, cout) and residual.is_contiguous())
    out = _new((h, w, n, cout) if out_hwnc else (n, h, w, cout), x)
    nb = 4.0 * (x.numel() + out.numel() * (2 if residual is not None else 1) + w_packed.numel())
    _run("conv3x3_fw32", 2.0 * out.numel() * 9 * cin, nb,
         lambda: _hip.lib().ssad_conv3x3_fw_eval(_hip.ptr(x), _hip.ptr(w_packed), _hip.ptr(out), _hip.ptr(shift), _hip.ptr(residual, True),
                                                 int(relu), n, h, w, cin, cout, int(out_hwnc), _hip.stream()))
    return out


def conv3x3_c64_eval(x, w_ohwi, scale=None, shift=None, residual=None, relu=False, in_hwnc=False, out_hwnc=False,
                     res_hwnc=None):
    """Halo-tile 3x3 / stride 1 / pad 1 convolution 64 -> 64 with the inference epilogue act(conv * scale + shift + residual).
    x is NHWC [N][H][W][64] or, with in_hwnc, position-major [H][W][N][64]; the output likewise under out_hwnc and the
    residual under res_hwnc (default: the output's layout)."""
    res_hwnc = out_hwnc if res_hwnc is None else res_hwnc
    if in_hwnc:
        h, w, n, c = x.shape
    else:
        n, h, w, c = x.shape
    assert c == 64 and tuple(w_ohwi.shape) == (64, 3, 3, 64)
    out = _new((h, w, n, 64) if out_hwnc else (n, h, w, 64), x)
    assert residual is None or tuple(residual.shape) == ((h, w, n, 64) if res_hwnc else (n, h, w, 64))
    nb = 4.0 * (x.numel() + out.numel() * (2 if residual is not None else 1) + w_ohwi.numel())
    _run("conv_c64_f32", 2.0 * out.numel() * 9 * 64, nb,
         lambda: _hip.lib().ssad_conv3x3_c64_eval(_hip.ptr(x), _hip.ptr(w_ohwi), _hip.ptr(out), _hip.ptr(scale, True),
                                                  _hip.ptr(shift, True), _hip.ptr(residual, True), int(relu), n, h, w,
                                                  int(in_hwnc), int(out_hwnc), int(res_hwnc), _hip.stream()))
    return out


def linear_fwd(x, w, scale=None, shift=None, relu=False, x3=False):
    """x [N][Cin], w [Cout][Cin] -> [N][Cout] (the same MFMA kernel with H=W=KH=KW=1)."""
    n, cin = x.shape
    cout = w.shape[0]
    out = _new((n, cout), x)
    if x3 and cout % 4 == 0:
        _run("conv_igemm_x6" if x3 == 6 else "conv_igemm_x3", 2.0 * n * cin * cout, 4.0 * (x.numel() + out.numel() + w.numel()),
             lambda: _split_fwd(x3)(_hip.ptr(x), _hip.ptr(w), _hip.ptr(out), _hip.ptr(scale, True),
                                    _hip.ptr(shift, True), None, int(relu), n, 1, 1, cin, cout, 1, 1, 1,
                                    0, 0, _hip.stream()))
        return out
    _run("conv_igemm_f32", 2.0 * n * cin * cout, 4.0 * (x.numel() + out.numel() + w.numel()),
         lambda: _hip.lib().ssad_conv_igemm_fwd(_hip.ptr(x), _hip.ptr(w), _hip.ptr(out), _hip.ptr(scale, True),
                                                _hip.ptr(shift, True), None, int(relu), n, 1, 1, cin, cout, 1, 1, 1, 0,
                                                _hip.stream()),
         tile=lambda: igemm_tile_name(n, 1, 1, cin, cout, 1, 1, 1, 0, 0))
    return out


def gap_fwd(x, out, offset, hwnc=False):
    if hwnc:
        h, w, n, c = x.shape
    else:
        n, h, w, c = x.shape
    if _is_h(x):
        assert not hwnc
        _run_aside("gap_h16", 0.0, 2.0 * x.numel() + 4.0 * n * c,
                   lambda: _hip.lib().ssad_gap_fwd_h(x.data_ptr(), _hip.ptr(out), n, h * w, c, out.shape[1], offset, _hip.stream()), keep=(x,))
        return out
    _run_aside("gap", 0.0, 4.0 * (x.numel() + n * c),
               lambda: _hip.lib().ssad_gap_fwd(_hip.ptr(x), _hip.ptr(out), n, h * w, c, out.shape[1], offset, int(hwnc),
                                               _hip.stream()), keep=(x,))
    return out


def l2_normalize_rows(x):
    out = torch.empty_like(x)
    _run("l2norm_rows", 0.0, 8.0 * x.numel(),
         lambda: _hip.lib().ssad_l2_normalize_rows(_hip.ptr(x), _hip.ptr(out), x.shape[0], x.shape[1], _hip.stream()))
    return out


def cosine_knn_mean(sim, k=3):
    out = _new((sim.shape[0],), sim)
    _run("knn_mean", 0.0, 4.0 * sim.numel(),
         lambda: _hip.lib().ssad_cosine_knn_mean(_hip.ptr(sim), _hip.ptr(out), sim.shape[0], sim.shape[1], k, _hip.stream()))
    return out


def cosine_knn_fused(x, bank_n, k=3):
    """x [N][D] (not normalised), bank_n [R][D] L2-normalised -> [N] mean of the k smallest cosine distances, one kernel."""
    n, d = x.shape
    r = bank_n.shape[0]
    out = _new((n,), x)
    _run("knn_fused", 2.0 * n * d * r, 4.0 * (x.numel() + bank_n.numel() + n),
         lambda: _hip.lib().ssad_cosine_knn_fused(_hip.ptr(x), _hip.ptr(bank_n), _hip.ptr(out), n, d, r, k, _hip.stream()))
    return out


def blur_relu_bilinear(maps, ksize=7, target=256):
    n, c, h, w = maps.shape
    out = _new((n, c, target, target), maps)
    _run("blur_relu_bilinear", 0.0, 4.0 * (maps.numel() + out.numel()),
         lambda: _hip.lib().ssad_blur_relu_bilinear(_hip.ptr(maps), _hip.ptr(out), n * c, h, w, ksize, target, _hip.stream()))
    return out


_RESAMPLE_TABLES = {}


def _resample_tables(n_in, n_out, device):
    """Device copies of Pillow's per-axis resampling tables, one pair per (input extent, output extent, device)."""
    key = (n_in, n_out, str(device))
    t = _RESAMPLE_TABLES.get(key)
    if t is None:
        from .pil_exact import resample_coeffs
        ks, bounds, kk = resample_coeffs(n_in, n_out)
        t = _RESAMPLE_TABLES[key] = (ks, torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device))
    return t


def resize_bicubic_u8(img, size):
    """Pillow's ``Image.resize(size)`` (BICUBIC) on a uint8 batch [B][Hin][Win][C] (C = 1: mode 'L', 3: 'RGB') that lives on the
    device; size = (width, height) as PIL takes it.  Bit-exact (csrc/resize.hip); same size = the input itself, as PIL copies."""
    b, hin, win, c = img.shape
    wout, hout = int(size[0]), int(size[1])
    if (hin, win) == (hout, wout):
        return img
    if b == 0:
        return torch.empty((0, hout, wout, c), dtype=torch.uint8, device=img.device)
    assert img.dtype == torch.uint8 and img.is_cuda and img.is_contiguous()
    out = torch.empty((b, hout, wout, c), dtype=torch.uint8, device=img.device)
    kx = ky = (0, None, None)
    if win != wout:
        kx = _resample_tables(win, wout, img.device)
    if hin != hout:
        ky = _resample_tables(hin, hout, img.device)
    tmp = torch.empty((b, hin, wout, c), dtype=torch.uint8, device=img.device) if (win != wout and hin != hout) else None
    p = lambda t: None if t is None else t.data_ptr()
    _run("resize_bicubic", 0.0, float(img.numel() + out.numel() + (2 * tmp.numel() if tmp is not None else 0)),
         lambda: _hip.lib().ssad_resize_bicubic_u8(img.data_ptr(), p(tmp), out.data_ptr(), b, hin, win, c, hout, wout,
                                                   p(kx[1]), p(kx[2]), kx[0], p(ky[1]), p(ky[2]), ky[0], _hip.stream()))
    return out


def obj_mask_batch(rgb, sigma=1.5, low=5, high=15, chunk=64, return_edges=False):
    """dataset_generator.obj_mask for a uint8 RGB batch [B][H][W][3] on the device -> bool [B][H][W] (csrc/objmask.hip; bit-exact
    against the host statement).  The Gaussian weights are scipy.ndimage's (numpy's exp on the host), everything per pixel runs on
    the device.  return_edges: also the Canny edge maps."""
    import ctypes
    import numpy as np
    assert rgb.dtype == torch.uint8 and rgb.is_cuda and rgb.is_contiguous() and rgb.shape[-1] == 3
    b, h, w, _ = rgb.shape
    radius = int(4.0 * float(sigma) + 0.5)                       # scipy.ndimage.gaussian_filter1d: truncate = 4
    xs = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * xs ** 2)
    phi = (phi / phi.sum())[::-1].copy()
    wts = (ctypes.c_double * len(phi))(*phi.tolist())
    lib = _hip.lib()
    mask = torch.empty((b, h, w), dtype=torch.uint8, device=rgb.device)
    edges = torch.empty((b, h, w), dtype=torch.uint8, device=rgb.device)
    for s in range(0, b, chunk):
        e = min(b, s + chunk)
        ws = torch.empty(lib.ssad_obj_mask_workspace(e - s, h, w), dtype=torch.uint8, device=rgb.device)
        _hip.check(lib.ssad_obj_mask(rgb[s:e].data_ptr(), mask[s:e].data_ptr(), edges[s:e].data_ptr(), e - s, h, w, wts, radius,
                                     low / 255.0, high / 255.0, ws.data_ptr(), _hip.stream()))
        torch.cuda.current_stream().synchronize()               # the weights live in this frame: the upload must have happened
    return (mask.bool(), edges.bool()) if return_edges else mask.bool()


def gradcam_map(act, alpha):
    """act NHWC [B][U][V][C], alpha [B][C] (may be a column slice of a wider matrix) -> [B][1][U][V] weighted sums."""
    b, u, v, c = act.shape
    assert alpha.shape == (b, c) and alpha.stride(1) == 1
    out = _new((b, 1, u, v), act)
    _run("gradcam_map", 2.0 * act.numel(), 4.0 * act.numel(),
         lambda: _hip.lib().ssad_gradcam_map(_hip.ptr(act), _hip.ptr(alpha), _hip.ptr(out), b, u * v, c, alpha.stride(0),
                                             _hip.stream()))
    return out


# ---------------------------------------------------------------------------------------------
# training kernels
# ---------------------------------------------------------------------------------------------
def flip_transpose_weight(w_ohwi):
    o, kh, kw, i = w_ohwi.shape
    out = _new((i, kh, kw, o), w_ohwi)
    _hip.check(_hip.lib().ssad_flip_transpose_weight(_hip.ptr(w_ohwi), _hip.ptr(out), o, i, kh, kw, _hip.stream()))
    return out


def conv_dgrad(dy, w_flipT, x_shape, stride, pad, residual=None, bf16=False, res_mask=None):
    """dy NHWC [N][Hy][Wy][Cout]; w_flipT [Cin][KH][KW][Cout] -> dx NHWC of x_shape (+ residual [under res_mask])."""
    n, hy, wy, cout = dy.shape
    cin, kh, kw, _ = w_flipT.shape
    assert residual is None or tuple(residual.shape) == tuple(x_shape), "residual must have dx's shape"
    assert tuple(x_shape) == (n, x_shape[1], x_shape[2], cin) and w_flipT.shape[-1] == cout
    if _is_h(dy):
        assert _is_h(w_flipT) and bf16 == 2 and res_mask is None and (residual is None or _is_h(residual))
        dx = _newh(tuple(x_shape), dy)
        _run("conv_igemm_h16", 2.0 * dx.numel() * kh * kw * cout / (stride * stride),
             2.0 * (dy.numel() + dx.numel() * (2 if residual is not None else 1) + w_flipT.numel()),
             lambda: _hip.lib().ssad_conv_igemm_dgrad_h(dy.data_ptr(), w_flipT.data_ptr(), dx.data_ptr(), _p(residual, True), n, hy, wy,
                                                        cout, x_shape[1], x_shape[2], cin, kh, kw, stride, pad, _hip.stream()))
        return dx
    dx = _new(tuple(x_shape), dy)
    if res_mask is not None:
        assert not bf16 and residual is not None
        _run("conv_igemm_f32", 2.0 * dx.numel() * kh * kw * cout / (stride * stride),
             4.0 * (dy.numel() + dx.numel() * 2 + w_flipT.numel()),
             lambda: _hip.lib().ssad_conv_igemm_dgrad_masked(_hip.ptr(dy), _hip.ptr(w_flipT), _hip.ptr(dx), _hip.ptr(residual),
                                                             res_mask.data_ptr(), n, hy, wy, cout, x_shape[1], x_shape[2], cin, kh,
                                                             kw, stride, pad, _hip.stream()),
             tile=lambda: igemm_tile_name(n, x_shape[1], x_shape[2], cout, cin, kh, kw, 1, kh - 1 - pad, 0).replace(
                 ",1,false>", f",{stride},false>"))
        return dx
    fn = (_hip.lib().ssad_conv_igemm_dgrad_x6 if bf16 == 6 else _hip.lib().ssad_conv_igemm_dgrad_x3 if bf16 == 3 else
          _hip.lib().ssad_conv_igemm_dgrad_f16 if bf16 == 2 else
          _hip.lib().ssad_conv_igemm_dgrad_bf16 if bf16 else _hip.lib().ssad_conv_igemm_dgrad)
    _run(_kname("conv_igemm", bf16),
         2.0 * dx.numel() * kh * kw * cout / (stride * stride),
         4.0 * (dy.numel() + dx.numel() * (2 if residual is not None else 1) + w_flipT.numel()),
         lambda: fn(_hip.ptr(dy), _hip.ptr(w_flipT), _hip.ptr(dx), _hip.ptr(residual, True), n, hy, wy, cout, x_shape[1],
                    x_shape[2], cin, kh, kw, stride, pad, _hip.stream()),
         tile=None if bf16 else (lambda: igemm_tile_name(n, x_shape[1], x_shape[2], cout, cin, kh, kw, 1, kh - 1 - pad, 0).replace(
             ",1,false>", f",{stride},false>")))
    return dx


def conv_wgrad(dy, x, dw_out, kh, kw, stride, pad, kreal=None, to_oihw=False, accumulate=False, bf16=False, force_x6=False):
    """dy NHWC [N][Ho][Wo][Cout], x NHWC [N][H][W][Cin] -> dw_out (flat, Cout*KH*KW*Cin_real floats).
    kreal = (KH, KW, Cin) of the real filter when x rows are padded im2col rows (stem)."""
    n, h, w, cin = x.shape
    cout = dy.shape[-1]
    m = dy.numel() // cout
    lib = _hip.lib()
    assert tuple(dy.shape[:-1]) == (n, (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1), \
        f"dy {tuple(dy.shape)} is not the output of a {kh} x {kw} / stride {stride} / pad {pad} conv over x {tuple(x.shape)}"
    if bf16 == 6 and not force_x6:
        bf16 = False            # (see below)
    if _is_h(dy):
        # precision-16 step with half tensors: the same two 16-bit kernels, their operands read as the halves they are stored as
        assert _is_h(x) and bf16 == 2 and kreal is None
        if lib.ssad_wgrad3x3_g16_ok(cin, cout, kh, kw, stride, pad):
            # 3 x 3 / pad 1, stride 1 or 2: tiles staged as they lie in memory, transposed by the fragment reads (csrc/wgrad16.hip)
            ho, wo = dy.shape[1], dy.shape[2]
            splits = lib.ssad_wgrad3x3_g16_splits(n, ho, wo, cin, cout, stride)
            slab = _new((splits, cout, 9 * cin), dy)
            _run("wgrad_g16", 2.0 * m * cout * 9 * cin, 2.0 * (dy.numel() + x.numel()) + 4.0 * slab.numel(),
                 lambda: lib.ssad_conv_wgrad3x3_g16_h(dy.data_ptr(), x.data_ptr(), _hip.ptr(slab), splits, n, ho, wo, h, w, cin, cout,
                                                      stride, dy.numel(), _hip.stream()))
        elif lib.ssad_wgrad3x3_halo16_ok(cin, cout, kh, kw, stride, pad):
            splits = lib.ssad_wgrad3x3_halo16_splits(n, h, w, cin, cout)
            slab = _new((splits, cout, 9 * cin), dy)
            _run("wgrad_h16", 2.0 * m * cout * 9 * cin, 2.0 * (dy.numel() + x.numel()) + 4.0 * slab.numel(),
                 lambda: lib.ssad_conv_wgrad3x3_halo16_h(dy.data_ptr(), x.data_ptr(), _hip.ptr(slab), splits, n, h, w, cin, cout,
                                                         dy.numel(), _hip.stream()))
        else:
            splits = lib.ssad_wgrad_splits_bf16(m, cin, cout, kh, kw)
            slab = _new((splits, cout, kh * kw * cin), dy)
            _run("wgrad_h16", 2.0 * m * cout * kh * kw * cin, 2.0 * (dy.numel() + x.numel()) * kh * kw + 4.0 * slab.numel(),
                 lambda: lib.ssad_conv_wgrad_f16_h(dy.data_ptr(), x.data_ptr(), _hip.ptr(slab), splits, n, h, w, cin, cout, kh, kw,
                                                   stride, pad, dy.numel(), _hip.stream()))
        _wgrad_reduce(slab, dw_out, splits, cout, kh * kw * cin, kh, kw, cin, to_oihw, accumulate)
        return dw_out
    if (int(bf16) in (0, 1, 2) and not _is_h(dy) and kreal is None and kh == 1 and kw == 1 and h == 1 and w == 1
            and m <= lib.ssad_linear_small_max_rows()):
        # linear layer over a training batch's rows: one launch straight into the gradient (OIHW == OHWI for 1 x 1); the 16-bit modes
        # round the operands while they are loaded
        _run_aside("wgrad_f32" if not bf16 else "wgrad_small16", 2.0 * m * cout * cin, 4.0 * (dy.numel() + x.numel() + cout * cin),
                   lambda: lib.ssad_linear_wgrad_small_r(_hip.ptr(dy), _hip.ptr(x), _hip.ptr(dw_out), m, cin, cout, int(accumulate), int(bf16),
                                                         _hip.stream()), keep=(dy, x))
        return dw_out
    halo = (lib.ssad_wgrad3x3_halo_ok(cin, cout, kh, kw, stride, pad)
            if not bf16 and kreal is None and os.environ.get("SSAD_WGRAD_HALO", "1") != "0" else 0)
    if halo:
        # 3x3 / pad 1 on the exact fp32 path: halo-tile kernel (one workgroup = a 64 x 64 block, all nine taps); stride 1 or 2
        ho, wo = dy.shape[1], dy.shape[2]
        assert (ho, wo) == ((h - 1) // stride + 1, (w - 1) // stride + 1)
        splits = lib.ssad_wgrad3x3_halo_splits(n, ho, wo, cin, cout)
        slab = _new((splits, cout, 9 * cin), dy)
        if halo == 1:
            fn = lambda: lib.ssad_conv_wgrad3x3_halo(_hip.ptr(dy), _hip.ptr(x), _hip.ptr(slab), splits, n, h, w, cin, cout,
                                                     dy.numel(), _hip.stream())
        else:
            fn = lambda: lib.ssad_conv_wgrad3x3s2_halo(_hip.ptr(dy), _hip.ptr(x), _hip.ptr(slab), splits, n, ho, wo, h, w, cin,
                                                       cout, dy.numel(), _hip.stream())
        _run("wgrad_f32", 2.0 * m * cout * 9 * cin, 4.0 * (dy.numel() + x.numel() + slab.numel()), fn)
        _wgrad_reduce(slab, dw_out, splits, cout, 9 * cin, 3, 3, cin, to_oihw, accumulate)
        return dw_out
    if bf16 in (1, 2, True) and kreal is None and lib.ssad_wgrad3x3_halo16_ok(cin, cout, kh, kw, stride, pad):
        # 16-bit operands, 3x3 / stride 1 / pad 1: halo-tile kernel (dZ and X fetched, converted and transposed once per pixel tile
        # instead of once per filter tap; csrc/wgrad_halo16.hip)
        splits = lib.ssad_wgrad3x3_halo16_splits(n, h, w, cin, cout)
        slab = _new((splits, cout, 9 * cin), dy)
        _run(_kname("wgrad", bf16), 2.0 * m * cout * 9 * cin, 4.0 * (dy.numel() + x.numel() + slab.numel()),
             lambda: lib.ssad_conv_wgrad3x3_halo16(_hip.ptr(dy), _hip.ptr(x), _hip.ptr(slab), splits, n, h, w, cin, cout,
                                                   int(bf16 == 2), dy.numel(), _hip.stream()))
        _wgrad_reduce(slab, dw_out, splits, cout, 9 * cin, 3, 3, cin, to_oihw, accumulate)
        return dw_out
    if bf16 == 6 and not force_x6:
        # bf16x6 training keeps weight gradients on the exact fp32 kernel: the wave-specialised fp32 wgrad (110 TFLOP/s) is
        # as fast as the six-product bf16 form (measured), and exact; ssad_conv_wgrad_x6 stays available (tests)
        bf16 = False
    splits = (_hip.lib().ssad_wgrad_splits_bf16 if bf16 else _hip.lib().ssad_wgrad_splits)(m, cin, cout, kh, kw)
    slab = _new((splits, cout, kh * kw * cin), dy)
    fn = (_hip.lib().ssad_conv_wgrad_x6 if bf16 == 6 else _hip.lib().ssad_conv_wgrad_x3 if bf16 == 3 else
          _hip.lib().ssad_conv_wgrad_f16 if bf16 == 2 else
          _hip.lib().ssad_conv_wgrad_bf16 if bf16 else _hip.lib().ssad_conv_wgrad)
    _run(_kname("wgrad", bf16), 2.0 * m * cout * kh * kw * cin,
         4.0 * (dy.numel() * kh * kw + x.numel() * kh * kw + slab.numel()),
         lambda: fn(_hip.ptr(dy), _hip.ptr(x), _hip.ptr(slab), splits, n, h, w, cin, cout, kh, kw, stride, pad, dy.numel(),
                    _hip.stream()))
    rkh, rkw, rcin = kreal if kreal else (kh, kw, cin)
    _wgrad_reduce(slab, dw_out, splits, cout, kh * kw * cin, rkh, rkw, rcin, to_oihw, accumulate)
    return dw_out


def stem_wgrad(img, dz, dw_out, to_oihw=False, accumulate=False):
    """Weight gradient of the stem conv from the NCHW image and dz [B][Ho][Wo][64] -> dw_out (64*147 floats)."""
    b, c, h, w = img.shape
    _, _, _, ho, wo = stem_geometry(h, w, 0, 0)
    assert c == 3 and tuple(dz.shape) == (b, ho, wo, 64), f"dz {tuple(dz.shape)} does not belong to images {tuple(img.shape)}"
    lib = _hip.lib()
    ws = torch.empty(lib.ssad_stem_wgrad_workspace(b, h, w), device=img.device, dtype=torch.float32)
    if _is_h(dz):
        _run("stem_wgrad_h16", 2.0 * dz.numel() * 147, 2.0 * dz.numel() + 4.0 * img.numel() * 1.6,
             lambda: lib.ssad_stem_wgrad_h(_hip.ptr(img), dz.data_ptr(), _hip.ptr(dw_out), b, h, w, dz.numel(), int(to_oihw), int(accumulate),
                                           _hip.ptr(ws), _hip.stream()))
        return dw_out
    _run("stem_wgrad", 2.0 * dz.numel() * 147, 4.0 * (dz.numel() + img.numel() * 1.6),
         lambda: lib.ssad_stem_wgrad(_hip.ptr(img), _hip.ptr(dz), _hip.ptr(dw_out), b, h, w, dz.numel(), int(to_oihw), int(accumulate),
                                     _hip.ptr(ws), _hip.stream()))
    return dw_out


def stem_im2col(img, hv, wv):
    b, c, h, w = img.shape
    ho, wo = (hv - 1) // 2 + 1, (wv - 1) // 2 + 1
    col = _new((b, ho, wo, 160), img)
    _run("stem_im2col", 0.0, 4.0 * (img.numel() + col.numel()),
         lambda: _hip.lib().ssad_stem_im2col(_hip.ptr(img), _hip.ptr(col), b, h, w, hv, wv, _hip.stream()))
    return col


def pack_stem_weight_2d(w_oihw):
    out = _new((64, 1, 1, 160), w_oihw)
    _hip.check(_hip.lib().ssad_pack_stem_weight_2d(_hip.ptr(w_oihw), _hip.ptr(out), _hip.stream()))
    return out


def _colreduce_ws(r, c, like):
    n = _hip.lib().ssad_colreduce_workspace(r, c)
    return torch.empty(n, device=like.device, dtype=torch.float64)


def bn_stats(z, c, eps, momentum, running_mean, running_var):
    r = z.numel() // c
    mean, invstd = _new((c,), z), _new((c,), z)
    ws = _colreduce_ws(r, c, z)
    fn = _hip.lib().ssad_bn_stats_h if _is_h(z) else _hip.lib().ssad_bn_stats
    _run("bn_stats", 0.0, z.element_size() * z.numel(),
         lambda: fn(_p(z), r, c, eps, momentum, _hip.ptr(mean), _hip.ptr(invstd),
                                          _hip.ptr(running_mean, True), _hip.ptr(running_var, True), ws.data_ptr(),
                                          _hip.stream()))
    return mean, invstd


def bn_small_ok(rows, c):
    return bool(_hip.lib().ssad_bn_small_ok(rows, c))


def bn_small_fwd(z, gamma, beta, eps, momentum, running_mean, running_var, relu):
    """Train-mode BatchNorm over a few hundred rows in one launch (statistics, running statistics, apply).  -> (y, mean, invstd)"""
    c = z.shape[-1]
    r = z.numel() // c
    y, mean, invstd = torch.empty_like(z), _new((c,), z), _new((c,), z)
    _run("bn_small_fwd", 0.0, 4.0 * 3 * z.numel(),
         lambda: _hip.lib().ssad_bn_small_fwd(_hip.ptr(z), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(y), _hip.ptr(mean),
                                              _hip.ptr(invstd), _hip.ptr(running_mean, True), _hip.ptr(running_var, True), r, c,
                                              eps, momentum, int(relu), _hip.stream()))
    return y, mean, invstd


def bn_small_bwd(dy, z, mean, invstd, gamma, zmask_beta, dbeta, dgamma, dbias=None):
    """Its backward in one launch: dbeta, dgamma (and dbias = column sums of dz) written, dz returned."""
    c = z.shape[-1]
    r = z.numel() // c
    dz = torch.empty_like(dy)
    _run("bn_small_bwd", 0.0, 4.0 * 5 * z.numel(),
         lambda: _hip.lib().ssad_bn_small_bwd(_hip.ptr(dy), _hip.ptr(z), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma),
                                              _hip.ptr(zmask_beta, True), _hip.ptr(dbeta, True), _hip.ptr(dgamma, True),
                                              _hip.ptr(dbias, True), _hip.ptr(dz), r, c, _hip.stream()))
    return dz


def bn_apply_fwd(z, mean, invstd, gamma, beta, residual, relu):
    c = mean.numel()
    y = torch.empty_like(z)
    assert residual is None or (residual.dtype == z.dtype and residual.shape == z.shape), "residual must have z's shape"
    assert z.shape[-1] == c
    fn = _hip.lib().ssad_bn_apply_fwd_h if _is_h(z) else _hip.lib().ssad_bn_apply_fwd
    _run("bn_apply_fwd_h16" if _is_h(z) else "bn_apply_fwd", 0.0, z.element_size() * z.numel() * (3 if residual is not None else 2),
         lambda: fn(_p(z), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma), _hip.ptr(beta),
                    _p(residual, True), _p(y), z.numel() // c, c, int(relu), _hip.stream()))
    return y


def bn_bwd_reduce(dy, yact, z, mean, invstd, dbeta, dgamma, c):
    r = dy.numel() // c
    ws = _colreduce_ws(r, c, dy)
    assert all(t is None or (t.dtype == dy.dtype and t.shape == dy.shape) for t in (yact, z)), "dy, yact, z must share one shape"
    fn = _hip.lib().ssad_bn_bwd_reduce_h if _is_h(dy) else _hip.lib().ssad_bn_bwd_reduce
    _run("bn_bwd_reduce_h16" if _is_h(dy) else "bn_bwd_reduce", 0.0, dy.element_size() * dy.numel() * (1 + (yact is not None) + (z is not None)),
         lambda: fn(_p(dy), _p(yact, True), _p(z, True), _hip.ptr(mean, True),
                                               _hip.ptr(invstd, True), _hip.ptr(dbeta, True), _hip.ptr(dgamma, True), r, c,
                                               ws.data_ptr(), _hip.stream()))


def bn_apply_bwd(dy, yact, z, mean, invstd, gamma, dbeta, dgamma, want_dres, eval_mode=False):
    c = mean.numel()
    dz = torch.empty_like(dy)
    dres = torch.empty_like(dy) if want_dres else None
    assert all(t is None or (t.dtype == dy.dtype and t.shape == dy.shape) for t in (yact, z)), "dy, yact, z must share one shape"
    assert dy.shape[-1] == c
    fn = _hip.lib().ssad_bn_apply_bwd_h if _is_h(dy) else _hip.lib().ssad_bn_apply_bwd
    _run("bn_apply_bwd_h16" if _is_h(dy) else "bn_apply_bwd", 0.0, dy.element_size() * dy.numel() * (3 + (yact is not None) + want_dres),
         lambda: fn(_p(dy), _p(yact, True), _p(z, True), _hip.ptr(mean),
                    _hip.ptr(invstd), _hip.ptr(gamma), _hip.ptr(dbeta, True), _hip.ptr(dgamma, True),
                    _p(dz), _p(dres, True), dy.numel() // c, c, int(eval_mode), _hip.stream()))
    return dz, dres


def bn_bwd_zmask(dy, z, mean, invstd, gamma, beta, dbeta, dgamma):
    """BatchNorm(train) + ReLU backward with the mask recomputed from z (no residual): returns dz."""
    c = mean.numel()
    r = dy.numel() // c
    ws = _colreduce_ws(r, c, dy)
    lib = _hip.lib()
    assert z.dtype == dy.dtype and z.shape == dy.shape, "dy and z must share one shape"
    half = _is_h(dy)
    es = dy.element_size()
    f_red = lib.ssad_bn_bwd_reduce_zmask_h if half else lib.ssad_bn_bwd_reduce_zmask
    f_app = lib.ssad_bn_apply_bwd_zmask_h if half else lib.ssad_bn_apply_bwd_zmask
    _run("bn_bwd_reduce_h16" if half else "bn_bwd_reduce", 0.0, 2.0 * es * dy.numel(),
         lambda: f_red(_p(dy), _p(z), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma),
                       _hip.ptr(beta), _hip.ptr(dbeta), _hip.ptr(dgamma), r, c, ws.data_ptr(), _hip.stream()))
    dz = torch.empty_like(dy)
    _run("bn_apply_bwd_h16" if half else "bn_apply_bwd", 0.0, 3.0 * es * dy.numel(),
         lambda: f_app(_p(dy), _p(z), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma),
                       _hip.ptr(beta), _hip.ptr(dbeta), _hip.ptr(dgamma), _p(dz), r, c, _hip.stream()))
    return dz


def bn_relu_maxpool_fwd(z, mean, invstd, gamma, beta, winners=False):
    """Training stem tail: pooled = maxpool3x3s2(relu(bn(z))) + argmax slots, without storing the activation.  winners: also the raw z of
    each window's winner ([N][Ho][Wo][C]): pool_bn_relu_bwd then takes the BatchNorm reduction over the pooled tensors."""
    n, h, w, c = z.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    idx = torch.empty((n, ho, wo, c), device=z.device, dtype=torch.uint8)
    half = _is_h(z)
    out = _newh((n, ho, wo, c), z) if half else _new((n, ho, wo, c), z)
    zw = torch.empty_like(out) if winners else None
    lib = _hip.lib()
    eb = z.element_size()
    if winners:
        fn = lib.ssad_bn_relu_maxpool_fwd_win_h if half else lib.ssad_bn_relu_maxpool_fwd_win
        _run("maxpool3x3s2_h16" if half else "maxpool3x3s2", 0.0, eb * (z.numel() + 2.0 * out.numel()) + idx.numel(),
             lambda: fn(z.data_ptr(), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma), _hip.ptr(beta), out.data_ptr(), idx.data_ptr(),
                        zw.data_ptr(), n, h, w, c, _hip.stream()))
        return out, idx, zw
    fn = lib.ssad_bn_relu_maxpool_fwd_h if half else lib.ssad_bn_relu_maxpool_fwd
    _run("maxpool3x3s2_h16" if half else "maxpool3x3s2", 0.0, eb * (z.numel() + out.numel()) + idx.numel(),
         lambda: fn(z.data_ptr(), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma), _hip.ptr(beta), out.data_ptr(), idx.data_ptr(), n, h, w, c,
                    _hip.stream()))
    return out, idx


def pool_bn_relu_bwd(idx, dpool, z, mean, invstd, gamma, beta, dbeta, dgamma, zwin=None):
    """Training stem head of the backward pass: (idx, dpool, z) -> dz, filling dbeta / dgamma.  zwin (bn_relu_maxpool_fwd(winners=True)):
    the BatchNorm reduction runs over (dpool, zwin) -- a quarter of the rows, no pass over z -- and only the apply pass reads z."""
    n, h, w, c = z.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    assert tuple(dpool.shape) == (n, ho, wo, c) and tuple(idx.shape) == (n, ho, wo, c) and dpool.dtype == z.dtype, \
        f"pooled gradient {tuple(dpool.shape)} / slots {tuple(idx.shape)} do not belong to z {tuple(z.shape)}"
    dz = torch.empty_like(z)
    half = _is_h(z)
    lib = _hip.lib()
    eb = z.element_size()
    if zwin is not None:
        assert tuple(zwin.shape) == (n, ho, wo, c) and zwin.dtype == z.dtype and dpool.is_contiguous() and zwin.is_contiguous()
        r = n * ho * wo
        ws = _colreduce_ws(r, c, z)
        red = lib.ssad_bn_bwd_reduce_zmask_h if half else lib.ssad_bn_bwd_reduce_zmask
        _run("bn_bwd_reduce_h16" if half else "bn_bwd_reduce", 0.0, eb * 2.0 * dpool.numel(),
             lambda: red(dpool.data_ptr(), zwin.data_ptr(), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(dbeta),
                         _hip.ptr(dgamma), r, c, ws.data_ptr(), _hip.stream()))
        app = lib.ssad_pool_bn_relu_bwd_apply_h if half else lib.ssad_pool_bn_relu_bwd_apply
        _run("pool_bn_bwd_h16" if half else "pool_bn_bwd", 0.0, eb * (2.0 * z.numel() + dpool.numel()),
             lambda: app(idx.data_ptr(), dpool.data_ptr(), z.data_ptr(), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma), _hip.ptr(beta),
                         _hip.ptr(dbeta), _hip.ptr(dgamma), dz.data_ptr(), n, h, w, c, dpool.numel(), _hip.stream()))
        return dz
    ws = _colreduce_ws(n * h * w, c, z)
    fn = lib.ssad_pool_bn_relu_bwd_h if half else lib.ssad_pool_bn_relu_bwd
    _run("pool_bn_bwd_h16" if half else "pool_bn_bwd", 0.0, eb * (3.0 * z.numel() + 2.0 * dpool.numel()),
         lambda: fn(idx.data_ptr(), dpool.data_ptr(), z.data_ptr(), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma), _hip.ptr(beta),
                    _hip.ptr(dbeta), _hip.ptr(dgamma), dz.data_ptr(), n, h, w, c, dpool.numel(), ws.data_ptr(), _hip.stream()))
    return dz


def maxpool3x3s2_bwd(x, dy):
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    _run("maxpool_bwd", 0.0, 4.0 * (2 * x.numel() + dy.numel()),
         lambda: _hip.lib().ssad_maxpool3x3s2_bwd(_hip.ptr(x), _hip.ptr(dy), _hip.ptr(dx), n, h, w, c, dy.numel(), _hip.stream()))
    return dx


def maxpool3x3s2_fwd_idx(x):
    """Training forward: pooled output + uint8 argmax slots."""
    n, h, w, c = x.shape
    out = _new((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), x)
    idx = torch.empty(out.shape, dtype=torch.uint8, device=x.device)
    _run("maxpool3x3s2", 0.0, 4.0 * (x.numel() + out.numel()) + idx.numel(),
         lambda: _hip.lib().ssad_maxpool3x3s2_fwd_idx(_hip.ptr(x), _hip.ptr(out), idx.data_ptr(), n, h, w, c, _hip.stream()))
    return out, idx


def maxpool3x3s2_bwd_idx(idx, dy, x_shape):
    n, h, w, c = x_shape
    assert tuple(dy.shape) == (n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c) and idx.shape == dy.shape, "dy / idx do not belong to x_shape"
    dx = _new(tuple(x_shape), dy)
    _run("maxpool_bwd", 0.0, 4.0 * (dx.numel() + dy.numel()) + idx.numel(),
         lambda: _hip.lib().ssad_maxpool3x3s2_bwd_idx(idx.data_ptr(), _hip.ptr(dy), _hip.ptr(dx), n, h, w, c, dy.numel(),
                                                      _hip.stream()))
    return dx


def gap_bwd(dpooled, dy, offset, accumulate):
    n, h, w, c = dy.shape
    if _is_h(dy):
        _run("gap_bwd_h16", 0.0, 2.0 * dy.numel() * (2 if accumulate else 1),
             lambda: _hip.lib().ssad_gap_bwd_h(_hip.ptr(dpooled), dy.data_ptr(), n, h * w, c, dpooled.shape[1], offset,
                                               int(accumulate), _hip.stream()))
        return dy
    _run("gap_bwd", 0.0, 4.0 * dy.numel() * (2 if accumulate else 1),
         lambda: _hip.lib().ssad_gap_bwd(_hip.ptr(dpooled), _hip.ptr(dy), n, h * w, c, dpooled.shape[1], offset,
                                         int(accumulate), _hip.stream()))
    return dy


def softmax_ce(logits, labels, dlogits=None, grad_scale=1.0):
    b, c = logits.shape
    out = _new((2,), logits)
    ldd = dlogits.shape[1] if dlogits is not None else 0
    _hip.check(_hip.lib().ssad_softmax_ce(_hip.ptr(logits), _hip.ptr(labels, dtype=torch.int64), b, c, _hip.ptr(out),
                                          _hip.ptr(dlogits, True), ldd, grad_scale, _hip.stream()))
    return out


def sgd_step(p, g, m, lr, momentum, weight_decay, grad_scale=1.0):
    _run("sgd", 0.0, 20.0 * p.numel(),
         lambda: _hip.lib().ssad_sgd_step(_hip.ptr(p), _hip.ptr(g), _hip.ptr(m), p.numel(), lr, momentum, weight_decay,
                                          grad_scale, _hip.stream()))


def sgd_step_dev(p, g, m, hyper, scaler=None):
    """SGD with [lr, momentum, weight_decay, grad_scale] (and the optional loss-scaler state) read from device memory."""
    _run("sgd", 0.0, 20.0 * p.numel(),
         lambda: _hip.lib().ssad_sgd_step_dev(_hip.ptr(p), _hip.ptr(g), _hip.ptr(m), p.numel(), _hip.ptr(hyper),
                                              _hip.ptr(scaler, True), _hip.stream()))


def scale_by_loss_scale(x, scaler):
    _hip.check(_hip.lib().ssad_scale_by_loss_scale(_hip.ptr(x), x.numel(), _hip.ptr(scaler), _hip.stream()))


def check_finite(g, scaler):
    _run("check_finite", 0.0, 4.0 * g.numel(),
         lambda: _hip.lib().ssad_check_finite(_hip.ptr(g), g.numel(), _hip.ptr(scaler), _hip.stream()))


def loss_scaler_update(scaler, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
    _hip.check(_hip.lib().ssad_loss_scaler_update(_hip.ptr(scaler), growth_factor, backoff_factor, growth_interval,
                                                  _hip.stream()))


def bn_apply_fwd_mask(z, mean, invstd, gamma, beta, residual, relu):
    """bn_apply_fwd that also returns the final ReLU's active set as a nibble mask (uint8 [R][C/4])."""
    c = mean.numel()
    y = torch.empty_like(z)
    mask = torch.empty(z.numel() // 4, device=z.device, dtype=torch.uint8)
    if _is_h(z):                  # half tensors: two mask bytes per lane (8 channels)
        assert residual is None or _is_h(residual)
        _run("bn_apply_fwd_h16", 0.0, 2.0 * z.numel() * (3 if residual is not None else 2) + mask.numel(),
             lambda: _hip.lib().ssad_bn_apply_fwd_mask_h(z.data_ptr(), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma), _hip.ptr(beta),
                                                         _p(residual, True), y.data_ptr(), mask.data_ptr(), z.numel() // c, c,
                                                         int(relu), _hip.stream()))
        return y, mask
    _run("bn_apply_fwd", 0.0, 4.0 * z.numel() * (3 if residual is not None else 2) + mask.numel(),
         lambda: _hip.lib().ssad_bn_apply_fwd_mask(_hip.ptr(z), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma), _hip.ptr(beta),
                                                   _hip.ptr(residual, True), _hip.ptr(y), mask.data_ptr(), z.numel() // c, c,
                                                   int(relu), _hip.stream()))
    return y, mask


def bn_bwd_mask(dy, mask, z, mean, invstd, gamma, dbeta, dgamma):
    """BatchNorm backward with g = dy * mask (mask None: g = dy): fills dbeta / dgamma, returns dz.  The identity-branch
    gradient (g itself) is not materialised: consumers apply `mask` to dy (conv_dgrad / conv3x3_c64 res_mask)."""
    c = mean.numel()
    r = dy.numel() // c
    ws = _colreduce_ws(r, c, dy)
    lib = _hip.lib()
    mp = mask.data_ptr() if mask is not None else None
    if _is_h(dy):
        assert _is_h(z)
        _run("bn_bwd_reduce_h16", 0.0, 4.0 * dy.numel() + (mask.numel() if mask is not None else 0),
             lambda: lib.ssad_bn_bwd_reduce_mask_h(dy.data_ptr(), mp, z.data_ptr(), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(dbeta),
                                                   _hip.ptr(dgamma), r, c, ws.data_ptr(), _hip.stream()))
        dz = torch.empty_like(dy)
        _run("bn_apply_bwd_h16", 0.0, 6.0 * dy.numel() + (mask.numel() if mask is not None else 0),
             lambda: lib.ssad_bn_apply_bwd_mask_h(dy.data_ptr(), mp, z.data_ptr(), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma),
                                                  _hip.ptr(dbeta), _hip.ptr(dgamma), dz.data_ptr(), r, c, _hip.stream()))
        return dz
    _run("bn_bwd_reduce", 0.0, 8.0 * dy.numel() + (mask.numel() if mask is not None else 0),
         lambda: lib.ssad_bn_bwd_reduce_mask(_hip.ptr(dy), mp, _hip.ptr(z), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(dbeta),
                                             _hip.ptr(dgamma), r, c, ws.data_ptr(), _hip.stream()))
    dz = torch.empty_like(dy)
    _run("bn_apply_bwd", 0.0, 12.0 * dy.numel() + (mask.numel() if mask is not None else 0),
         lambda: lib.ssad_bn_apply_bwd_mask(_hip.ptr(dy), mp, _hip.ptr(z), _hip.ptr(mean), _hip.ptr(invstd), _hip.ptr(gamma),
                                            _hip.ptr(dbeta), _hip.ptr(dgamma), _hip.ptr(dz), r, c, _hip.stream()))
    return dz


def cvt_f32_f16(src, dst):
    """One rounded (half) copy of a flat fp32 arena: the weights the half-tensor kernels of the precision-16 step read."""
    assert src.dtype == torch.float32 and dst.dtype == torch.float16 and src.numel() == dst.numel()
    _run("cvt_f32_f16", 0.0, 6.0 * src.numel(),
         lambda: _hip.lib().ssad_cvt_f32_f16(_hip.ptr(src), dst.data_ptr(), src.numel(), _hip.stream()))
    return dst
