"""Forward engine: runs PeraNet's trunk + head on the HIP kernels (NHWC, fp32 MFMA).

Mirrors the op sequence of ``PeraNet.forward`` (src/self_supervised/models.py:210-253 of the
reference): [patch window + nearest resize +] stem, max-pool, 8 BasicBlocks, global average pool of
layer2/3/4 concatenated in that order, concatenator, latent MLP, classifier.  Eval-mode BatchNorm is
folded to a per-channel scale/shift applied in the conv epilogue (alpha = gamma/sqrt(var+eps),
beta' = beta - mean*alpha: the same two-constant form PyTorch's CPU kernel evaluates).
"""
import torch
from torch import nn

import os as _os

from . import ops

BLOCKS = [("layer1", 64, 64, 1), ("layer2", 64, 128, 2), ("layer3", 128, 256, 2), ("layer4", 256, 512, 2)]


def math_mode():
    """Product arithmetic of the eval-mode trunk / head, read per call (tests and bench.py switch it): SSAD_MATH =
    "f32" (default: exact fp32 MFMA), "bf16x6" (three-way bf16 split, six MFMAs: fp32-faithful products, 1.2e-6 against
    fp64 where the fp32 MFMA shows 1.5e-6, ~1.4x the conv throughput) or "bf16x3" (two-way split, three MFMAs: 4.6e-6,
    2.1-2.4x).  Returns 0, 6 or 3."""
    m = _os.environ.get("SSAD_MATH", "f32").lower()
    return 6 if m in ("bf16x6", "x6") else 3 if m in ("bf16x3", "x3") else 0


class _Block(nn.Module):
    """Parameter holder with torchvision BasicBlock names; never called."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        self.stride = stride


class ResNet18Params(nn.Module):
    """Holds the trunk's parameters/buffers under torchvision's state_dict names (SURVEY s.5).
    Weights are random (kaiming-normal fan_out, as torchvision initialises): the IMAGENET1K_V1
    checkpoint the reference downloads (models.py:59) is loaded via load_state_dict where available."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        for name, cin, cout, stride in BLOCKS:
            setattr(self, name, nn.Sequential(_Block(cin, cout, stride), _Block(cout, cout, 1)))
        if self.conv1.weight.is_meta:          # shapes only (PeraNet.load_from_checkpoint): the checkpoint supplies every tensor
            return
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, *a, **k):
        raise RuntimeError("ResNet18Params only holds parameters; PeraNet.forward drives the HIP engine")


def _fold_bn(bn, bias=None):
    """eval BN -> (scale, shift); an optional preceding Linear bias is folded into the shift."""
    with torch.no_grad():
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
        if bias is not None:
            shift = shift + bias * scale
    return scale.contiguous(), shift.contiguous()


class EvalPlan:
    """Packed (OHWI / stem-order) weights and folded BN constants for one parameter version."""

    def __init__(self, model):
        fe = model.feature_extractor
        self.version = param_version(model)
        with torch.no_grad():
            self.stem_w = ops.pack_stem_weight(fe.conv1.weight.contiguous())
            self.stem_wf = ops.pack_stem_weight_folded(fe.conv1.weight.contiguous())
            self.stem_s, self.stem_t = _fold_bn(fe.bn1)
            self.blocks = []
            for name, _, _, _ in BLOCKS:
                for blk in getattr(fe, name):
                    d = {"stride": blk.stride,
                         "w1": ops.repack_oihw_to_ohwi(blk.conv1.weight.contiguous()),
                         "w2": ops.repack_oihw_to_ohwi(blk.conv2.weight.contiguous())}
                    d["s1"], d["t1"] = _fold_bn(blk.bn1)
                    d["s2"], d["t2"] = _fold_bn(blk.bn2)
                    if blk.downsample is not None:
                        d["wd"] = ops.repack_oihw_to_ohwi(blk.downsample[0].weight.contiguous())
                        d["sd"], d["td"] = _fold_bn(blk.downsample[1])
                    self.blocks.append((name, d))
            self.head = []
            lin, bn = model.concatenator[0], model.concatenator[1]
            self.head.append((lin.weight.contiguous(), *_fold_bn(bn, lin.bias), False))
            ls = list(model.latent_space)
            for seq in ls[:-2]:
                self.head.append((seq[0].weight.contiguous(), *_fold_bn(seq[1], seq[0].bias), True))
            self.head.append((ls[-2].weight.contiguous(), *_fold_bn(ls[-1], ls[-2].bias), False))
            self.cls_w = model.classifier.weight.contiguous()
            self.cls_b = model.classifier.bias.contiguous()


def param_version(model):
    return tuple(t._version for t in list(model.parameters()) + list(model.buffers())) + (
        next(model.parameters()).device,)


def trunk_eval(plan, x, patch_dim, patch_stride, layer_outputs, pooled):
    """x NCHW fp32 -> writes the GAP vectors of the requested stages into ``pooled`` [N][D].

    Many samples with small maps (the 841-patches-per-image scoring path: 64x64 inputs, maps 16x16 .. 2x2) run in
    the position-major layout [H][W][N][C]: a conv workgroup then owns 128 samples at one output position, reads
    contiguous rows for every tap and skips the taps that fall into the zero padding (8 % .. 56 % of the MACs)."""
    b, _, h, w = x.shape
    p, hv, wv, _, _ = ops.stem_geometry(h, w, patch_dim, patch_stride)
    hwnc = b * p >= 128 and hv * wv <= 64 * 64
    x3 = math_mode()
    if hwnc:
        conv = (lambda *a: ops.conv_fwd_hwnc(*a, x3=x3)) if x3 else ops.conv_fwd_hwnc
    else:
        conv = (lambda *a: ops.conv_fwd(*a, x3)) if x3 else ops.conv_fwd
    c64 = not x3 and _os.environ.get("SSAD_C64_EVAL", "1") != "0"      # exact-fp32 layer1 through csrc/conv_c64.hip
    # ... or csrc/conv16w.hip's inference form (SSAD_CONV32W_EVAL=1).  OFF: measured on the scoring pass (8 launches of 107 648 maps of
    # 16 x 16 x 64, round 5) 128.6 ms against 125.1 for the c64 kernel -- 64 input channels give tiles of four 16-channel chunks, and the
    # residual costs fp32 MFMAs there
    fw_eval = c64 and _os.environ.get("SSAD_CONV32W_EVAL", "0") == "1"
    win = (patch_dim, patch_dim) if patch_dim else (h, w)
    # Overlapping patches share layer1's arithmetic (round 6, DESIGN s.4): windows of 32 pixels at an even stride -> the pooled maps of
    # neighbouring patches are shifts of ONE per-image map by stride / 2 positions, and a conv output whose receptive field stays
    # clear of the patch's own zero-padded border equals the conv of that dense map.  Exact fp32 only (SSAD_DEDUP=0 switches it off).
    dedup = (hwnc and patch_dim == 32 and not x3 and patch_stride % 2 == 0 and h >= 64 and w >= 64
             and _os.environ.get("SSAD_DEDUP", "1") != "0")
    if dedup:
        return _trunk_eval_dedup(plan, x, patch_stride, layer_outputs, pooled, conv)
    if win == (32, 32):
        # exact 2x nearest upsample: folded 4x4 conv + BN + ReLU + max-pool fused, the conv map never reaches HBM
        a = ops.stem_patch_pool_fwd(x, plan.stem_wf, plan.stem_s, plan.stem_t, patch_stride if patch_dim else 1, hwnc and not c64)
    else:
        a = ops.stem_fwd(x, plan.stem_w, plan.stem_s, plan.stem_t, True, patch_dim, patch_stride, hwnc and not c64)
        a = ops.maxpool3x3s2_fwd(a, hwnc and not c64)
    offs, off = {}, 0
    for k in ("layer1", "layer2", "layer3"):
        if k in layer_outputs:
            offs[k] = off
            off += {"layer1": 64, "layer2": 128, "layer3": 256}[k]
    offs["layer4"] = off
    for i, (name, d) in enumerate(plan.blocks):
        s = d["stride"]
        idt = a
        if "wd" in d:
            idt = conv(a, d["wd"], d["sd"], d["td"], None, False, s, 0)
        if c64 and name == "layer1":
            # halo-tile kernel: one halo load per 8 x 16 pixel tile instead of one gather per filter tap
            # (measured at 15 979 patches of 16 x 16: 2.39 ms against 2.53 position-major implicit GEMM; NHWC tensors are
            # another 5 % faster than position-major ones, so layer1 stays NHWC and its last conv writes [H][W][N][C])
            if fw_eval and a.dim() == 4 and ops.conv3x3_fw_eval_ok(a.shape[0], a.shape[1], a.shape[2], 64, 64):
                # launches that fill the chip: the register-fed conv (csrc/conv16w.hip, T = float), BatchNorm scale folded into its
                # packed filter once per plan, two 16 x 16 maps per tile
                if "p1" not in d:
                    d["p1"], d["p2"] = ops.conv3x3_fw_pack_scaled(d["w1"], d["s1"]), ops.conv3x3_fw_pack_scaled(d["w2"], d["s2"])
                t = ops.conv3x3_fw_eval(a, d["p1"], 64, d["t1"], None, True, False)
                a = ops.conv3x3_fw_eval(t, d["p2"], 64, d["t2"], idt, True, hwnc and i == 1)
            else:
                t = ops.conv3x3_c64_eval(a, d["w1"], d["s1"], d["t1"], None, True, False, False)
                a = ops.conv3x3_c64_eval(t, d["w2"], d["s2"], d["t2"], idt, True, False, hwnc and i == 1, False)
        else:
            t = conv(a, d["w1"], d["s1"], d["t1"], None, True, s, 1)
        if not (c64 and name == "layer1"):
            a = conv(t, d["w2"], d["s2"], d["t2"], idt, True, 1, 1)
        last_of_stage = (i % 2 == 1)
        if last_of_stage and name in offs:
            ops.gap_fwd(a, pooled, offs[name], hwnc)
    return pooled


def _trunk_eval_dedup(plan, x, patch_stride, layer_outputs, pooled, conv):
    """trunk_eval for 32 x 32 windows: layer1 computed once per IMAGE wherever overlapping patches agree.

    A patch (pr, pc) is the window [ps pr : ps pr + 32) of the image, nearest-upsampled 2x (models.py:217-219), so its 64 x 64 input
    is a window of the image's 2x upsample at offset 2 ps, its 32 x 32 stem map a window of the image's stem map at offset ps and
    its 16 x 16 pooled map a window of the image's pooled map at offset ps / 2 -- EXCEPT where the patch's own zero padding is
    within reach: stem rows 0, 1, 31 (7 x 7 / 2, pad 3), hence pooled rows 0, 1, 15, and one more row per side for every 3 x 3 conv
    that follows.  So after conv j of layer1 (j = 1 .. 4) the positions 2 + j <= u, v <= 14 - j of every patch equal the dense map
    (11^2, 9^2, 7^2, 5^2 of 256 positions): they are copied (ssad_patch_gather_hwnc), the ring around them is computed patch-wise
    (ssad_conv_igemm_fwd_hwnc_ring), and the dense maps cost 1 / 13 of a patch-wise conv.  Every product that is computed is the
    exact fp32 product the patch-wise conv forms; only the summation order inside a position differs between the two kernels."""
    b, _, h, w = x.shape
    prow, pcol = (h - 32) // patch_stride + 1, (w - 32) // patch_stride + 1
    shift = patch_stride // 2
    band = _os.environ.get("SSAD_GATHER_BAND", "1") != "0"
    # (the pooled map's interior [4, 12]^2 is read by nobody: the first ring conv computes the outputs outside [3, 13] and reads within
    # one position of them, block 0's second conv takes its residual at the positions outside [4, 12])
    dn = ops.stem_fwd(x, plan.stem_w, plan.stem_s, plan.stem_t, True, resize_to=(2 * h, 2 * w))         # [B][h][w][64]
    dn = ops.maxpool3x3s2_fwd(dn)                                                                       # [B][h/2][w/2][64]
    if band and _os.environ.get("SSAD_STEM_BORDER", "1") != "0":
        # the pooled map itself: only rows / columns 0, 1, 15 are the patch's own (13 of the stem's 32 tiles of matrix work); rows /
        # columns 2-3 and 13-14 are the per-image pooled map's
        a = ops.stem_patch_border_fwd(x, plan.stem_wf, plan.stem_s, plan.stem_t, patch_stride)          # [16][16][N][64]
        ops.patch_gather_hwnc(dn, a, prow, pcol, shift, 2, 14, 4, 12)
    else:
        a = ops.stem_patch_pool_fwd(x, plan.stem_wf, plan.stem_s, plan.stem_t, patch_stride, True, (4, 12) if band else None)
    offs, off = {}, 0
    for k in ("layer1", "layer2", "layer3"):
        if k in layer_outputs:
            offs[k] = off
            off += {"layer1": 64, "layer2": 128, "layer3": 256}[k]
    offs["layer4"] = off
    for i, (name, d) in enumerate(plan.blocks):
        s = d["stride"]
        if name == "layer1":
            lo, hi = 3 + 2 * i, 13 - 2 * i                       # interior after this block's first conv; one less per side after its second
            dt = ops.conv3x3_c64_eval(dn, d["w1"], d["s1"], d["t1"], None, True)
            dn2 = ops.conv3x3_c64_eval(dt, d["w2"], d["s2"], d["t2"], dn, True)
            # (the copies leave out what nobody reads: a ring conv reads its input within one position of the outputs it computes, so
            # of t only the positions outside [lo + 2, hi - 2] are read, of block 0's output only those outside [lo + 3, hi - 3] -- by
            # block 1's first conv and as the residual of its second; block 1's output feeds layer2 whole)
            t = ops.conv_fwd_hwnc_ring(a, d["w1"], d["s1"], d["t1"], None, True, lo, hi)
            ops.patch_gather_hwnc(dt, t, prow, pcol, shift, lo, hi, *((lo + 2, hi - 2) if band else (1, 0)))
            a2 = ops.conv_fwd_hwnc_ring(t, d["w2"], d["s2"], d["t2"], a, True, lo + 1, hi - 1)
            ops.patch_gather_hwnc(dn2, a2, prow, pcol, shift, lo + 1, hi - 1, *((lo + 3, hi - 3) if band and i == 0 else (1, 0)))
            a, dn = a2, dn2
        else:
            idt = a
            if "wd" in d:
                idt = conv(a, d["wd"], d["sd"], d["td"], None, False, s, 0)
            t = conv(a, d["w1"], d["s1"], d["t1"], None, True, s, 1)
            a = conv(t, d["w2"], d["s2"], d["t2"], idt, True, 1, 1)
        if i % 2 == 1 and name in offs:
            ops.gap_fwd(a, pooled, offs[name], True)
    return pooled


def head_eval(plan, pooled):
    f = pooled
    x3 = math_mode()
    for w, s, t, relu in plan.head:
        f = ops.linear_fwd(f, w, s, t, relu, x3)
    logits = ops.linear_fwd(f, plan.cls_w, None, plan.cls_b, False, x3)
    return logits, f
