"""Oracle: PeraNet forward / train step on torch-CPU fp32.

Follows src/self_supervised/models.py:58-146 (structure), :210-253 (forward),
:256-308 (train / val step), :336-341 (optimiser) of the reference.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .resnet18 import ResNet18
from .scoring import extract_patches


def _mlp(dim=512, n_hidden=3):
    # models.py:65-88 -- (latent_space_layers-1)=4 entries: 3 x [Linear(nb)+BN+ReLU],
    # then Linear(bias)+BN1d.  All widths are 512 (quirk Q9 is harmless).
    layers = [nn.Sequential(nn.Linear(dim, dim, bias=False), nn.BatchNorm1d(dim), nn.ReLU(inplace=True))
              for _ in range(n_hidden)]
    layers += [nn.Linear(dim, dim, bias=True), nn.BatchNorm1d(dim)]
    return nn.Sequential(*layers)


class OraclePeraNet(nn.Module):
    """state_dict keys identical to the reference's PeraNet (153 entries, SURVEY s.5)."""

    def __init__(self, layer_outputs=("layer2", "layer3"), num_classes=4, latent_space_layers=5):
        super().__init__()
        self.feature_extractor = ResNet18()
        self.feature_extractor.fc = nn.Identity()          # models.py:60-61
        self.layer_outputs = tuple(layer_outputs)
        dim = 512 + (64 if "layer1" in layer_outputs else 0) \
            + (128 if "layer2" in layer_outputs else 0) + (256 if "layer3" in layer_outputs else 0)
        self.concatenator = nn.Sequential(nn.Linear(dim, 512, bias=False), nn.BatchNorm1d(512))  # :91-95
        # models.py:138-141 hands latent_space_layers - 1 to the builder of :65-88: that many entries, all but the last hidden
        self.latent_space = _mlp(n_hidden=max(latent_space_layers - 1, 1) - 1)
        self.classifier = nn.Linear(512, num_classes)       # :98-99
        self.patch_level = False
        self.batch = None
        self.num_patches = None

    def trunk_features(self, x):
        """Returns dict of per-stage NCHW activations (what the forward hooks capture, :110-130)."""
        fe = self.feature_extractor
        x = fe.maxpool(fe.relu(fe.bn1(fe.conv1(x))))
        acts = {}
        for i in range(1, 5):
            x = getattr(fe, f"layer{i}")(x)
            acts[f"layer{i}"] = x
        return acts

    def forward(self, x):
        if self.patch_level:                                 # models.py:211-216
            x = extract_patches(x, dim=32, stride=8)
            b, p, c, h, w = x.shape
            x = x.reshape(b * p, c, h, w)
            self.batch, self.num_patches = b, p
        if x.shape[2] < 64 or x.shape[3] < 64:               # :217-219
            x = F.interpolate(x, 64, mode="nearest")
        acts = self.trunk_features(x)
        gap = lambda t: torch.flatten(F.adaptive_avg_pool2d(t, (1, 1)), 1)
        feats = [gap(acts[k]) for k in ("layer1", "layer2", "layer3") if k in self.layer_outputs]
        feats.append(gap(acts["layer4"]))                     # concat order l1|l2|l3|l4, :240-245
        features = torch.cat(feats, dim=1)
        features = self.concatenator(features)
        embeddings = self.latent_space(features)
        y_hat = self.classifier(embeddings)
        return {"classifier": y_hat, "latent_space": embeddings, "pooled": torch.cat(feats, dim=1)}


def train_step(model, x, y):
    """models.py:256-263: forward, cross-entropy, accuracy.  Returns (loss, acc, outputs)."""
    out = model(x)
    loss = F.cross_entropy(out["classifier"], y)
    acc = (out["classifier"].argmax(1) == y).float().mean()
    return loss, acc, out


def make_optimizer(model, lr, epochs, stage):
    """models.py:336-341: SGD(m=.9, wd=5e-4); cosine warm restarts only when fine tuning."""
    opt = torch.optim.SGD(model.parameters(), lr, momentum=0.9, weight_decay=0.0005)
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, epochs) if stage == "fine_tune" else None
    return opt, sched


def cosine_warm_restart_lr(base_lr, epoch, t0, eta_min=0.0):
    """Closed form of CosineAnnealingWarmRestarts(T_0=t0, T_mult=1) stepped once per epoch."""
    t_cur = epoch % t0
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / t0)) / 2
