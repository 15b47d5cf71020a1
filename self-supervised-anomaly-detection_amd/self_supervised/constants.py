"""Containers and category lists (mirrors src/self_supervised/constants.py:7-118 of the reference)."""
from __future__ import annotations

import torch
from torch import Tensor

_FIELDS = ("original_data", "tensor_data", "y_true_binary_labels", "raw_predictions", "y_hat",
           "y_true_multiclass_labels", "ground_truths", "anomaly_maps", "embedding_vectors")
_OPTIONAL_IN_CAT = ("ground_truths", "anomaly_maps")


class ModelOutputsContainer:
    """The 9 tensor fields ``predict_step`` / ``tools.inference`` hand around (constants.py:7-53)."""

    def __init__(self) -> None:
        for f in _FIELDS:
            setattr(self, f, None)

    def to_cpu(self):
        for f in _FIELDS:
            v = getattr(self, f)
            setattr(self, f, v.to('cpu') if torch.is_tensor(v) else None)

    def from_list(self, predictions: list[ModelOutputsContainer]):
        cols = {f: [] for f in _FIELDS}
        for p in predictions:
            p.to_cpu()
            for f in _FIELDS:
                v = getattr(p, f)
                if f in _OPTIONAL_IN_CAT and not torch.is_tensor(v):
                    continue
                cols[f].append(v)
        for f in _FIELDS:
            setattr(self, f, torch.cat(cols[f]) if len(cols[f]) else None)


    def split(self, n_images: int) -> list:
        """One container per image: the inverse of ``from_list`` for a container that holds ``n_images`` images
        (patch-level fields carry num_patches rows per image, image-level fields one)."""
        parts = []
        for i in range(n_images):
            c = ModelOutputsContainer()
            for f in _FIELDS:
                v = getattr(self, f)
                if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] and v.shape[0] % n_images == 0:
                    k = v.shape[0] // n_images
                    setattr(c, f, v[i * k:(i + 1) * k].clone())
                else:
                    setattr(c, f, v)
            parts.append(c)
        return parts


class EvaluationOutputContainer:
    def __init__(self) -> None:
        self.auroc = None
        self.f1_score = None
        self.aupro = None
        self.iou = None

    def to_string(self) -> str:
        r = lambda v: round(v, 2) if v else None
        return ("scores: [\n    auroc: {0},\n    f1-score: {1},\n    aupro: {2},\n    iou: {3}\n]"
                .format(r(self.auroc), r(self.f1_score), r(self.aupro), r(self.iou)))


def METRICS() -> list:
    return ['auroc', 'f1-score', 'aupro', 'iou']


def TEXTURES() -> list:
    return ['carpet', 'grid', 'leather', 'tile', 'wood']


def OBJECTS() -> list:
    return ['bottle', 'cable', 'capsule', 'hazelnut', 'metal_nut', 'pill', 'screw', 'tile', 'toothbrush',
            'transistor', 'zipper']


def OBJECTS_SET_ONE() -> list:
    return ['bottle', 'cable', 'capsule', 'hazelnut', 'metal_nut']


def OBJECTS_SET_TWO() -> list:
    return ['pill', 'screw', 'toothbrush', 'transistor', 'zipper']


def NON_FIXED_OBJECTS() -> list:
    return ['hazelnut', 'screw', 'metal_nut']
