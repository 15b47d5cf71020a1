"""Tensor-level wrappers over the C ABI: allocate outputs with torch, launch on the current stream."""
import torch

from . import _hip


def _new(shape, like):
    return torch.empty(shape, device=like.device, dtype=torch.float32)


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
    assert tuple(w.shape) == (64, 3, 7, 7)
    out = _new((168, 64), w)
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


def stem_fwd(img, wk, scale, shift, relu=True, patch_dim=0, patch_stride=0, out=None):
    b, c, h, w = img.shape
    assert c == 3
    p, hv, wv, ho, wo = stem_geometry(h, w, patch_dim, patch_stride)
    if out is None:
        out = _new((b * p, ho, wo, 64), img)
    _hip.check(_hip.lib().ssad_stem_fwd(_hip.ptr(img), b, h, w, patch_dim, patch_stride, hv, wv, _hip.ptr(wk),
                                        _hip.ptr(scale, True), _hip.ptr(shift, True), int(relu), _hip.ptr(out),
                                        _hip.stream()))
    return out


def maxpool3x3s2_fwd(x):
    n, h, w, c = x.shape
    out = _new((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), x)
    _hip.check(_hip.lib().ssad_maxpool3x3s2_fwd(_hip.ptr(x), _hip.ptr(out), n, h, w, c, _hip.stream()))
    return out


def conv_fwd(x, w_ohwi, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0):
    """x NHWC [N][H][W][Cin]; w OHWI [Cout][KH][KW][Cin] -> NHWC [N][Ho][Wo][Cout]."""
    n, h, w, cin = x.shape
    cout, kh, kw, cin2 = w_ohwi.shape
    assert cin == cin2
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    out = _new((n, ho, wo, cout), x)
    _hip.check(_hip.lib().ssad_conv_igemm_fwd(_hip.ptr(x), _hip.ptr(w_ohwi), _hip.ptr(out), _hip.ptr(scale, True),
                                              _hip.ptr(shift, True), _hip.ptr(residual, True), int(relu), n, h, w, cin,
                                              cout, kh, kw, stride, pad, _hip.stream()))
    return out


def linear_fwd(x, w, scale=None, shift=None, relu=False):
    """x [N][Cin], w [Cout][Cin] -> [N][Cout] (the same MFMA kernel with H=W=KH=KW=1)."""
    n, cin = x.shape
    cout = w.shape[0]
    out = _new((n, cout), x)
    _hip.check(_hip.lib().ssad_conv_igemm_fwd(_hip.ptr(x), _hip.ptr(w), _hip.ptr(out), _hip.ptr(scale, True),
                                              _hip.ptr(shift, True), None, int(relu), n, 1, 1, cin, cout, 1, 1, 1, 0,
                                              _hip.stream()))
    return out


def gap_fwd(x, out, offset):
    n, h, w, c = x.shape
    _hip.check(_hip.lib().ssad_gap_fwd(_hip.ptr(x), _hip.ptr(out), n, h * w, c, out.shape[1], offset, _hip.stream()))
    return out


def l2_normalize_rows(x):
    out = torch.empty_like(x)
    _hip.check(_hip.lib().ssad_l2_normalize_rows(_hip.ptr(x), _hip.ptr(out), x.shape[0], x.shape[1], _hip.stream()))
    return out


def cosine_knn_mean(sim, k=3):
    out = _new((sim.shape[0],), sim)
    _hip.check(_hip.lib().ssad_cosine_knn_mean(_hip.ptr(sim), _hip.ptr(out), sim.shape[0], sim.shape[1], k, _hip.stream()))
    return out


def blur_relu_bilinear(maps, ksize=7, target=256):
    n, c, h, w = maps.shape
    out = _new((n, c, target, target), maps)
    _hip.check(_hip.lib().ssad_blur_relu_bilinear(_hip.ptr(maps), _hip.ptr(out), n * c, h, w, ksize, target, _hip.stream()))
    return out
