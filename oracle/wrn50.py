"""Oracle for BASELINE.json configs[3]: WideResNet-50-2 layer1-3 features + per-scale cosine 3-NN distance maps.

NO REFERENCE COUNTERPART: the reference hard-wires resnet18 (src/self_supervised/models.py:58-62) and has no multi-scale
feature-distance scorer; this config is listed in BASELINE.json as a throughput case only.  What is restated here:
torchvision's ``wide_resnet50_2`` (Bottleneck, v1.5: the stride sits on the 3x3 conv; width = 2 x planes; expansion 4) from
the public spec, up to layer3, and a scorer built from the reference's own pieces -- cosine 3-NN mean against a bank
(models.py:345-370) per scale, then relu(gaussian_blur(k = 7)) + bilinear to the input size (tools.py:394-399) and the mean over
the three scales.  Test infrastructure only (tests/, bench.py cpu_baseline)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import scoring as osc


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride):
        super().__init__()
        width = planes * 2                                  # wide_resnet50_2: width_per_group = 128
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class WideResNet50Trunk(nn.Module):
    """conv7x7/2 - bn - relu - maxpool3x3/2 - layer1 (3 blocks, 256 ch) - layer2 (4, 512, /2) - layer3 (6, 1024, /2)."""
    LAYERS = (("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 6, 2))

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for name, planes, blocks, stride in self.LAYERS:
            mods = []
            for b in range(blocks):
                mods.append(Bottleneck(cin, planes, stride if b == 0 else 1))
                cin = planes * 4
            setattr(self, name, nn.Sequential(*mods))

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        feats = []
        for name, _, _, _ in self.LAYERS:
            x = getattr(self, name)(x)
            feats.append(x)
        return feats                                        # NCHW: (N,256,H/4,W/4), (N,512,H/8,W/8), (N,1024,H/16,W/16)


def seeded_trunk(seed=0):
    """Random-init weights of the architecture (kaiming-normal convs, as torchvision initialises) with non-trivial BatchNorm
    statistics and scales, so that activations stay O(1) through 13 blocks."""
    g = torch.Generator().manual_seed(seed)
    m = WideResNet50Trunk()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.Conv2d):
                fan_out = mod.weight.shape[0] * mod.weight.shape[2] * mod.weight.shape[3]
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (2.0 / fan_out) ** 0.5)
            if isinstance(mod, nn.BatchNorm2d):
                mod.running_mean.copy_(0.1 * torch.randn(mod.num_features, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.num_features, generator=g))
                mod.weight.copy_(0.5 + 0.5 * torch.rand(mod.num_features, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.num_features, generator=g))
    return m.eval()


def seeded_banks(rows=588, seed=2):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(rows, c, generator=g) for c in (256, 512, 1024)]


def distance_maps(feats, banks, size, k=3):
    """feats: NCHW per scale; banks: [rows][C] per scale -> (N,1,size,size): mean over scales of
    bilinear(relu(blur7(cosine k-NN mean map)))."""
    out = None
    for f, b in zip(feats, banks):
        n, c, h, w = f.shape
        rows = f.permute(0, 2, 3, 1).reshape(-1, c).numpy()
        s, _, _ = osc.cosine_knn_mean(b.numpy(), rows, k)
        up = osc.upsample(torch.from_numpy(s).reshape(n, 1, h, w), size)
        out = up if out is None else out + up
    return out / float(len(feats))
