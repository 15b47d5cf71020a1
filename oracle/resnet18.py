"""Oracle: ResNet-18 trunk, restated from the public torchvision spec.

Restated, third-party: the reference calls ``torchvision.models.resnet18``
(src/self_supervised/models.py:58-62) and torchvision is not vendored under
/root/reference nor installed here.  The layers are plain ``torch.nn`` ops that
exist in this image, so the restatement is checkable op by op on CPU.
``state_dict`` keys match torchvision's (conv1, bn1, layer{1-4}.{0,1}.*,
downsample.{0,1}.*) so reference checkpoints map one to one.
"""
import torch
from torch import nn


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        if self.downsample is not None:
            idt = self.downsample(x)
        y = y + idt
        return self.relu(y)


class ResNet18(nn.Module):
    """conv7x7/2 - bn - relu - maxpool3x3/2 - 4 stages of 2 BasicBlocks - avgpool - fc."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        chans = [64, 64, 128, 256, 512]
        for i in range(1, 5):
            stride = 1 if i == 1 else 2
            setattr(self, f"layer{i}", nn.Sequential(
                BasicBlock(chans[i - 1], chans[i], stride),
                BasicBlock(chans[i], chans[i], 1)))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


def resnet18(weights=None, **kw):
    """Signature-compatible stand-in for ``torchvision.models.resnet18``.

    ``weights`` is accepted and ignored: the ImageNet checkpoint is not
    available offline; callers load seeded weights afterwards.
    """
    return ResNet18()
