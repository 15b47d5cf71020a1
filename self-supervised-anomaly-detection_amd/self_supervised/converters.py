"""Label helpers used by predict_step (mirrors src/self_supervised/converters.py:7-12)."""
import torch
from torch import Tensor


def gt2label(gt_list: Tensor, negative: int = 0, positive: int = 1) -> list:
    """One label per ground-truth mask: ``negative`` when the mask is empty."""
    sums = torch.as_tensor(gt_list).reshape(len(gt_list), -1).sum(dim=1)
    return [negative if s == 0 else positive for s in sums.tolist()]


def multiclass2binary(labels: Tensor) -> Tensor:
    return (torch.as_tensor(labels) > 0).to(torch.int64).cpu()
