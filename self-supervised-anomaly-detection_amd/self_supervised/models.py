"""PeraNet and AnomalyDetector with the reference's call surface, computed by HIP kernels.

Drop-in for src/self_supervised/models.py of gabry1998/Self-Supervised-Anomaly-Detection:
same constructor arguments, method names, return types and state_dict keys.  Numerics run in
libssad_hip.so (see engine.py / ops.py); there is no CPU fallback -- inputs must live on the GPU.
"""
import os
import warnings
from collections import OrderedDict

import numpy as np
import torch
from torch import Tensor, nn

from . import engine, ops
from .constants import ModelOutputsContainer
from .converters import gt2label, multiclass2binary
from .functional import get_prediction_class

try:                                    # optional: behave as a LightningModule when PL is installed
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:                       # PL is absent in the build image; trainer.py supplies the loop
    pl = None

    class _Base(nn.Module):
        def __init__(self):
            super().__init__()
            self.current_epoch = 0
            self.trainer = None
            self.hparams = {}
            self.logged = OrderedDict()

        def save_hyperparameters(self, **kw):
            self.hparams = dict(kw)

        def log_dict(self, metrics, **kw):
            for k, v in metrics.items():
                self.logged.setdefault(k, []).append(float(v))


_WARNED_RANDOM_BACKBONE = False


def _warn_random_backbone():
    global _WARNED_RANDOM_BACKBONE
    if not _WARNED_RANDOM_BACKBONE:
        _WARNED_RANDOM_BACKBONE = True
        warnings.warn("PeraNet: no ImageNet resnet18 weights found ($SSAD_RESNET18_WEIGHTS or torch hub cache "
                      "resnet18-f37072fd.pth); the backbone is RANDOMLY initialised.  The reference loads "
                      "IMAGENET1K_V1 (models.py:59): load a checkpoint / call load_backbone() before training.",
                      RuntimeWarning, stacklevel=3)


class PeraNet(_Base):
    """src/self_supervised/models.py:21-341."""

    def __init__(self, learning_rate: float = 0.03, epochs: int = 30, layer_outputs: list = ['layer2', 'layer3'],
                 latent_space_layers: int = 5, latent_space_layers_base_dim: int = 512, num_classes: int = 4,
                 memory_bank_dim: int = 1000, stage='projection_train') -> None:
        super().__init__()
        if pl is not None:
            self.save_hyperparameters()
        else:
            self.save_hyperparameters(learning_rate=learning_rate, epochs=epochs, layer_outputs=layer_outputs,
                                      latent_space_layers=latent_space_layers,
                                      latent_space_layers_base_dim=latent_space_layers_base_dim,
                                      num_classes=num_classes, memory_bank_dim=memory_bank_dim, stage=stage)
        self.backbone = 'resnet18'
        self.layer_outputs = list(layer_outputs)
        dim_in = 512 + sum({'layer1': 64, 'layer2': 128, 'layer3': 256}[k] for k in self.layer_outputs)
        base = latent_space_layers_base_dim
        self.feature_extractor = engine.ResNet18Params()
        self.concatenator = nn.Sequential(nn.Linear(dim_in, base, bias=False), nn.BatchNorm1d(base))
        # (latent_space_layers-1) entries: hidden [Linear(nb)+BN+ReLU] blocks then Linear(bias)+BN (models.py:65-88)
        n_hidden = max(latent_space_layers - 1, 1) - 1
        layers = [nn.Sequential(nn.Linear(base, base, bias=False), nn.BatchNorm1d(base), nn.ReLU(inplace=True))
                  for _ in range(n_hidden)]
        layers += [nn.Linear(base, 512, bias=True), nn.BatchNorm1d(512)]
        self.latent_space = nn.Sequential(*layers)
        self.classifier = nn.Linear(512, num_classes)

        self.mvtec = False
        self.patch_level = False
        self.num_classes = num_classes
        self.lr = learning_rate
        self.num_epochs = epochs
        self.stage = stage
        self.memory_bank_dim = memory_bank_dim
        self.memory_bank = torch.tensor([], device='cpu')
        self.batch = None
        self.num_patches = None
        # upper bound on the patches pushed through the trunk per kernel sequence; the bound actually used is derived per call
        # from the free HBM (_samples_per_pass: ~4 live layer1-sized tensors per sample, 60 % of what is free) and halves on an
        # out-of-memory error, so a smaller card, ranks sharing a device or a resident training graph pool get smaller passes
        # instead of failing.  Measured on an idle 288 GB card: 390.8 ms per 256 images at 16 384, 384.0 ms at 131 072, identical
        # results.  Every activation stays below 2^31 elements (131 072 x 16 x 16 x 64 would be exactly 2^31: hence the -1).
        self.max_samples_per_pass = 131072
        self.max_elements_per_tensor = 2 ** 31 - 1
        self.hbm_fraction_per_pass = 0.6
        self._plan = None
        self._frozen = set()
        # models.py:59 asks torchvision for IMAGENET1K_V1 (and fails loudly without it); there is no hub here, so the
        # same file is taken from $SSAD_RESNET18_WEIGHTS or torch's hub cache.  Without it the trunk keeps its random
        # init -- fine when a checkpoint / state dict is loaded next, useless for stage 1 of tools.training (frozen
        # backbone) -- so that case warns once per process (SSAD_ALLOW_RANDOM_BACKBONE=1 silences it).
        self.pretrained_backbone = False
        if self.classifier.weight.is_meta:          # load_from_checkpoint builds shapes only: the checkpoint supplies every tensor
            return
        for cand in (os.environ.get("SSAD_RESNET18_WEIGHTS"),
                     os.path.expanduser("~/.cache/torch/hub/checkpoints/resnet18-f37072fd.pth")):
            if cand and os.path.isfile(cand):
                self.load_backbone(cand)
                self.pretrained_backbone = True
                break
        if not self.pretrained_backbone and os.environ.get("SSAD_ALLOW_RANDOM_BACKBONE") != "1":
            _warn_random_backbone()

    def load_backbone(self, weights) -> None:
        """Load a torchvision ``resnet18`` state dict (or a path to one, e.g. resnet18-f37072fd.pth) into the trunk:
        what ``models.resnet18(weights="IMAGENET1K_V1")`` + ``fc = Identity`` leave behind (models.py:59-61)."""
        sd = torch.load(weights, map_location="cpu", weights_only=True) if isinstance(weights, (str, os.PathLike)) else weights
        sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
        own = self.feature_extractor.state_dict()
        missing = sorted(set(own) - set(sd))
        extra = sorted(set(sd) - set(own))
        if missing or extra:
            raise KeyError(f"not a resnet18 state dict: missing {missing[:4]}, unexpected {extra[:4]}")
        self.feature_extractor.load_state_dict(sd, strict=True)
        self._plan = None

    # ---- mode switches (models.py:149-172) ----
    def enable_patch_level_mode(self):
        self.patch_level = True

    def disable_patch_level_mode(self):
        self.patch_level = False

    def enable_mvtec_inference(self) -> None:
        self.mvtec = True

    def disable_mvtec_inference(self) -> None:
        self.mvtec = False

    def clear_memory_bank(self) -> None:
        self.memory_bank = torch.tensor([])

    def unfreeze_net(self, modules: list = ['backbone', 'latent_space']) -> None:
        if 'backbone' in modules:
            for p in self.feature_extractor.parameters():
                p.requires_grad = True
            self._frozen.discard('backbone')
        if 'latent_space' in modules:
            for m in (self.concatenator, self.latent_space):
                for p in m.parameters():
                    p.requires_grad = True
            self._frozen.discard('latent_space')

    def freeze_net(self, modules: list = ['backbone', 'latent_space']) -> None:
        if 'backbone' in modules:
            for p in self.feature_extractor.parameters():
                p.requires_grad = False
            self.feature_extractor.eval()
            self._frozen.add('backbone')
        if 'latent_space' in modules:
            for m in (self.concatenator, self.latent_space):
                for p in m.parameters():
                    p.requires_grad = False
                m.eval()
            self._frozen.add('latent_space')

    def unfreeze(self) -> None:          # LightningModule.unfreeze(), used at tools.py:282
        for p in self.parameters():
            p.requires_grad = True
        self._frozen.clear()
        self.train()

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location='cpu', **overrides):
        """LightningModule.load_from_checkpoint: rebuild from ``hyper_parameters`` (+ overrides, tools.py:277-281),
        load ``state_dict``, restore the memory bank.  Callable on the class or on an instance (quirk Q8)."""
        try:
            # memory-mapped: the tensors are read from the page cache when they move to the device, not copied into fresh host storage
            # first (the optimizer state of a full checkpoint is never touched at all)
            ck = torch.load(checkpoint_path, map_location=map_location, weights_only=False, mmap=True)
        except (RuntimeError, ValueError, TypeError):            # legacy (non-zipfile) checkpoints cannot be mapped
            ck = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
        hp = dict(ck.get('hyper_parameters', {}))
        hp.update(overrides)
        # every parameter and buffer comes from the checkpoint: build the module on the meta device (shapes only, no random
        # initialisation of 12.7 M weights: 0.2 s -> 0.03 s) and ADOPT the checkpoint's tensors
        with torch.device('meta'):
            model = cls(**hp)
        model.load_state_dict(ck['state_dict'], strict=True, assign=True)
        # (assign=True keeps every parameter's requires_grad flag)  Nothing may be left on the meta device: a non-persistent buffer or
        # a plain tensor attribute created in __init__ would otherwise fail at its first use, far from here
        left = [n for n, t in list(model.named_parameters()) + list(model.named_buffers()) if t.is_meta]
        if left:
            raise RuntimeError(f"load_from_checkpoint: {left} are not in the checkpoint's state_dict (still on the meta device)")
        model.on_load_checkpoint(ck)
        return model

    def on_save_checkpoint(self, checkpoint) -> None:
        checkpoint['memory_bank'] = self.memory_bank.to('cpu')

    def on_load_checkpoint(self, checkpoint) -> None:
        self.memory_bank = checkpoint['memory_bank'] if 'memory_bank' in checkpoint else torch.tensor([])

    # ---- forward (models.py:210-253) ----
    def _eval_plan(self):
        v = engine.param_version(self)
        if self._plan is None or self._plan.version != v:
            self._plan = engine.EvalPlan(self)
        return self._plan

    def _samples_per_pass(self, b, p, hv, wv, pd, device=None):
        """Images per trunk pass: the configured cap, fewer than 2^31 elements in the largest activation (the stem map: 1/4 of the
        network input's pixels x 64 channels per sample; the 32 x 32 patch path fuses stem + pool: 1/16), and what the free HBM
        holds (free = the driver's free bytes + what torch's allocator has cached but not handed out)."""
        shrink = 4 if pd == 32 else 2
        act = max(1, (hv // shrink) * (wv // shrink) * 64)                       # floats of the largest activation per sample
        cap = min(self.max_samples_per_pass, self.max_elements_per_tensor // act)
        free, _ = torch.cuda.mem_get_info(device)
        free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        live = 4 if pd == 32 else 3                                              # tensors of that size alive at once (+ 25 % for the deeper stages)
        cap = min(cap, int(self.hbm_fraction_per_pass * free / (act * 4 * live * 1.25)))
        per_pass = max(1, cap // p)
        return -(-b // -(-b // per_pass))               # equal passes (256 images: 2 x 128 rather than 155 + 101)

    def forward(self, x: Tensor) -> dict:
        if not x.is_cuda:
            raise RuntimeError("PeraNet.forward runs on the MI355X HIP kernels only: move the batch to the GPU")
        x = x.contiguous().float()
        if self.training and torch.is_grad_enabled():
            from . import training
            return training.forward_train(self, x)
        b, _, h, w = x.shape
        pd, ps = (32, 8) if self.patch_level else (0, 0)
        p, hv, wv = ops.stem_geometry(h, w, pd, ps)[:3]
        if self.patch_level:
            self.batch, self.num_patches = b, p
        plan = self._eval_plan()
        dim_in = self.concatenator[0].in_features
        pooled = torch.empty((b * p, dim_in), device=x.device, dtype=torch.float32)
        per_pass = self._samples_per_pass(b, p, hv, wv, pd, x.device)
        i0 = 0
        while i0 < b:
            i1 = min(b, i0 + per_pass)
            try:
                engine.trunk_eval(plan, x[i0:i1], pd, ps, self.layer_outputs, pooled[i0 * p:i1 * p])
            except torch.cuda.OutOfMemoryError:
                if per_pass == 1:
                    raise
                torch.cuda.empty_cache()
                per_pass = max(1, per_pass // 2)           # the estimate was too generous for what else lives on the card
                continue
            self.last_pass_samples = (i1 - i0) * p if i0 == 0 else self.last_pass_samples
            i0 = i1
        logits, emb = engine.head_eval(plan, pooled)
        return {'classifier': logits, 'latent_space': emb}

    # ---- steps (models.py:256-333) ----
    def training_step(self, batch, batch_idx) -> Tensor:
        from . import training
        return training.training_step(self, batch, batch_idx)

    def on_train_epoch_end(self) -> None:
        self.memory_bank = self.memory_bank[-self.memory_bank_dim:].clone()

    def fill_memory_bank(self, embeds: Tensor, y: Tensor, y_hat: Tensor):
        mask = (y == 0) & (y_hat == 0)
        embeds = embeds[mask].detach().to('cpu')
        self.memory_bank = torch.cat([self.memory_bank, embeds])[-self.memory_bank_dim:].clone()

    def validation_step(self, batch, batch_idx) -> dict:
        from . import training
        x, y, _ = batch
        with torch.no_grad():
            was = self.training
            self.eval()
            out = self(x)
            self.train(was)
        loss, acc = training.cross_entropy_eval(out['classifier'], y)
        metrics = {"val_accuracy": acc, "val_loss": loss}
        self.log_dict(metrics, on_step=False, on_epoch=True, prog_bar=True)
        return metrics

    def predict_step(self, batch, batch_idx, dataloader_idx=0) -> ModelOutputsContainer:
        outputs = ModelOutputsContainer()
        x_prime, groundtruths, x = batch
        if self.mvtec:
            outputs.y_true_binary_labels = torch.tensor(gt2label(groundtruths))
            outputs.y_true_multiclass_labels = torch.tensor(gt2label(groundtruths, negative=-1, positive=self.num_classes))
            outputs.ground_truths = groundtruths
        else:
            outputs.y_true_binary_labels = multiclass2binary(groundtruths)
            outputs.y_true_multiclass_labels = groundtruths
        with torch.no_grad():
            predictions = self(x_prime)
        raw_predictions = predictions['classifier']
        outputs.original_data = x
        outputs.tensor_data = x_prime
        outputs.raw_predictions = raw_predictions
        outputs.embedding_vectors = predictions['latent_space']
        outputs.y_hat = get_prediction_class(raw_predictions)
        return outputs

    def configure_optimizers(self):
        from . import training
        optimizer = training.FusedSGD(self, self.lr, momentum=0.9, weight_decay=0.0005)
        scheduler = training.CosineWarmRestarts(optimizer, self.num_epochs)
        if self.stage == 'fine_tune':
            return [optimizer], [scheduler]
        return [optimizer], []


def split_indices(n, test_size=0.3):
    """Index form of sklearn.model_selection.train_test_split(test_size=..., random_state=None, shuffle=True):
    n_test = ceil(test_size*n); one permutation from the global numpy RNG; test = its first n_test entries,
    train = the rest (quirk Q5: unseeded in the reference)."""
    n_test = int(np.ceil(test_size * n))
    perm = np.random.permutation(n)
    return perm[n_test:], perm[:n_test]


class AnomalyDetector:
    """src/self_supervised/models.py:345-370: cosine 3-NN distance to a bank of normal embeddings.

    ``fit`` keeps the reference's unseeded 70/30 split (quirk Q5: depends on the global numpy RNG);
    the bank is L2-normalised once and kept on the GPU, ``predict`` = normalise + MFMA GEMM + top-3 mean."""

    def __init__(self, patch_level: bool = False, batch: int = None, num_patches: int = None) -> None:
        self.patch_level = patch_level
        self.batch = batch
        self.dim = int(np.sqrt(num_patches)) if num_patches else None
        self.k = 3
        self.bank = None
        self.threshold = None

    @staticmethod
    def _dev(t):
        t = torch.as_tensor(t, dtype=torch.float32)
        if not t.is_cuda:
            if not torch.cuda.is_available():
                raise RuntimeError("AnomalyDetector needs the MI355X HIP kernels (no CPU fallback)")
            t = t.cuda()
        return t.contiguous()

    def fit(self, embeddings: Tensor, split: bool = True) -> None:
        emb = torch.as_tensor(embeddings)
        n = emb.shape[0]
        if split:
            train_idx, val_idx = split_indices(n, 0.3)
            train, val = emb[train_idx], emb[val_idx]
        else:
            train, val = emb, emb
        self.k = 3
        self.fit_bank(train)
        scores = self._scores(self._dev(val))
        self.threshold = torch.max(scores).item()

    def fit_bank(self, bank: Tensor) -> None:
        self.bank = ops.l2_normalize_rows(self._dev(bank))

    def _scores(self, x):
        if x.shape[1] % 32 == 0 and 1 <= self.k <= 3:
            # normalise + similarity GEMM + k smallest distances in one kernel: no N x bank matrix in HBM (csrc/knn.hip)
            return ops.cosine_knn_fused(x, self.bank, self.k)
        qn = ops.l2_normalize_rows(x)
        out = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
        step = 1 << 18
        for i in range(0, x.shape[0], step):
            sim = ops.linear_fwd(qn[i:i + step], self.bank)
            out[i:i + step] = ops.cosine_knn_mean(sim, self.k)
        return out

    def predict(self, x: Tensor) -> Tensor:
        anomaly_scores = self._scores(self._dev(x))
        if self.patch_level:
            anomaly_scores = torch.reshape(anomaly_scores, (self.batch, 1, self.dim, self.dim))
        return anomaly_scores
