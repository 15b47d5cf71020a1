"""WideResNet-50-2 multi-scale (layer1-3) feature-distance maps on the HIP kernels -- BASELINE.json configs[3].

No reference counterpart: the reference hard-wires resnet18 (src/self_supervised/models.py:58-62).  The config is a throughput
case built from the path's own kernels: the stem + max-pool of the ResNet trunk, Bottleneck blocks as three implicit-GEMM convs
(1x1 / 3x3 / 1x1 on the fp32 matrix cores, eval-mode BatchNorm + residual + ReLU folded into their epilogues), and per scale the
scorer of models.py:345-370 -- cosine 3-NN mean against a normality bank -- followed by tools.upsample (tools.py:394-399), averaged
over the three scales.  Parity: oracle/wrn50.py (torch-CPU restatement of the same definition)."""
import torch
from torch import nn

from . import engine, ops

LAYERS = (("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 6, 2))
FW_EVAL = __import__("os").environ.get("SSAD_WRN_FW", "1") != "0"      # 0: every conv on the implicit GEMM (rounds 3-5)


class _Bottleneck(nn.Module):
    """Parameter holder with torchvision's Bottleneck names; never called."""

    def __init__(self, cin, planes, stride):
        super().__init__()
        width = planes * 2
        self.conv1, self.bn1 = nn.Conv2d(cin, width, 1, bias=False), nn.BatchNorm2d(width)
        self.conv2, self.bn2 = nn.Conv2d(width, width, 3, stride, 1, bias=False), nn.BatchNorm2d(width)
        self.conv3, self.bn3 = nn.Conv2d(width, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride != 1 or cin != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        self.stride = stride


class WideResNet50Features(nn.Module):
    """wide_resnet50_2 up to layer3 under torchvision's state_dict names; ``forward`` returns the three NHWC feature maps."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for name, planes, blocks, stride in LAYERS:
            mods = []
            for b in range(blocks):
                mods.append(_Bottleneck(cin, planes, stride if b == 0 else 1))
                cin = planes * 4
            setattr(self, name, nn.Sequential(*mods))
        self._plan = None

    def _build_plan(self):
        with torch.no_grad():
            plan = {"stem_w": ops.pack_stem_weight(self.conv1.weight.contiguous()), "stem": engine._fold_bn(self.bn1), "blocks": []}
            for name, _, _, _ in LAYERS:
                for blk in getattr(self, name):
                    d = {"name": name, "stride": blk.stride}
                    for k in (1, 2, 3):
                        d[f"w{k}"] = ops.repack_oihw_to_ohwi(getattr(blk, f"conv{k}").weight.contiguous())
                        d[f"s{k}"], d[f"t{k}"] = engine._fold_bn(getattr(blk, f"bn{k}"))
                    if blk.stride == 1 and FW_EVAL:
                        # the 3 x 3 / stride 1 convs on the register-fed kernel (csrc/conv16w.hip, float form: 0.86-0.9 of the fp32-MFMA
                        # peak where the implicit GEMM reaches 0.81): BatchNorm scale folded into the packed filter, shift + ReLU in
                        # its epilogue
                        d["p2"] = ops.conv3x3_fw_pack_scaled(d["w2"], d["s2"])
                    if blk.downsample is not None:
                        d["wd"] = ops.repack_oihw_to_ohwi(blk.downsample[0].weight.contiguous())
                        d["sd"], d["td"] = engine._fold_bn(blk.downsample[1])
                    plan["blocks"].append(d)
        return plan

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("WideResNet50Features runs on the MI355X HIP kernels only: move the batch to the GPU")
        v = engine.param_version(self)
        if self._plan is None or self._plan[0] != v:
            self._plan = (v, self._build_plan())
        plan = self._plan[1]
        x = x.contiguous().float()
        a = ops.stem_fwd(x, plan["stem_w"], plan["stem"][0], plan["stem"][1], True)
        a = ops.maxpool3x3s2_fwd(a)
        feats, last = [], None
        for d in plan["blocks"]:
            if last is not None and d["name"] != last:
                feats.append(a)
            last = d["name"]
            idt = a if "wd" not in d else ops.conv_fwd(a, d["wd"], d["sd"], d["td"], None, False, d["stride"], 0)
            t = ops.conv_fwd(a, d["w1"], d["s1"], d["t1"], None, True, 1, 0)
            if "p2" in d and ops.conv3x3_fw_eval_ok(t.shape[0], t.shape[1], t.shape[2], t.shape[3], t.shape[3]):
                t = ops.conv3x3_fw_eval(t, d["p2"], t.shape[3], d["t2"], None, True)
            else:
                t = ops.conv_fwd(t, d["w2"], d["s2"], d["t2"], None, True, d["stride"], 1)
            a = ops.conv_fwd(t, d["w3"], d["s3"], d["t3"], idt, True, 1, 0)
        feats.append(a)
        return feats


class FeatureDistanceScorer:
    """Per-scale cosine k-NN mean against a bank of normal features (AnomalyDetector's scorer, models.py:345-370, applied to
    every pixel of every scale), tools.upsample per scale, mean over the scales."""

    def __init__(self, banks, k=3):
        self.k = k
        self.banks = [ops.l2_normalize_rows(b.contiguous().float()) for b in banks]

    def __call__(self, feats, size):
        out = None
        for f, bank in zip(feats, self.banks):
            n, h, w, c = f.shape
            score = ops.cosine_knn_fused(f.reshape(n * h * w, c), bank, self.k)
            up = ops.blur_relu_bilinear(score.view(n, 1, h, w), 7, size)
            out = up if out is None else out.add_(up)
        return out.div_(float(len(feats)))
