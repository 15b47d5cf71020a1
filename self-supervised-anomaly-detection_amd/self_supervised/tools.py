"""Pipelines with the reference's signatures (src/self_supervised/tools.py:204-399).

``upsample`` is the HIP blur+ReLU+bilinear kernel.  ``training`` / ``inference`` drive the built-in
fit/predict loop of trainer.py (PyTorch Lightning is not required)."""
import torch
from torch import Tensor

from . import ops


def upsample(anomaly_maps: Tensor, target_size: int = 256, verbose: bool = True):
    """tools.py:394-399: relu(gaussian_blur(k=7)) then bilinear to target_size, one fused kernel."""
    if verbose:
        print('>>> upsampling')
    m = torch.as_tensor(anomaly_maps, dtype=torch.float32)
    if not m.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("tools.upsample runs on the MI355X HIP kernel only (no CPU fallback)")
        m = m.cuda()
    return ops.blur_relu_bilinear(m.contiguous(), 7, target_size)
