"""Oracle: Grad-CAM saliency on torch-CPU fp32 (test infrastructure only).

Follows src/self_supervised/gradcam.py:25-48 of the reference: eval-mode forward, the gradient of one class logit with
respect to the layer4 output, alpha = spatial mean of that gradient, saliency = relu(sum_k alpha_k * A_k), bilinear
resize to the input size, min-max normalisation over the returned tensor.  The reference is only usable with one
image per call (score.backward() needs a scalar); `gradcam` keeps that contract, `gradcam_batch` loops it.
"""
import torch
import torch.nn.functional as F


def gradcam(model, x, class_idx=None):
    """x [1][3][H][W] -> saliency [1][1][H][W] in [0, 1] (NaN when the map is constant, as in the reference)."""
    assert x.shape[0] == 1
    model.eval()
    acts = {}
    handle = model.feature_extractor.layer4.register_forward_hook(lambda m, i, o: acts.__setitem__("value", o))
    try:
        out = model(x.clone())
    finally:
        handle.remove()
    logit = out["classifier"]
    idx = logit.max(1)[-1] if class_idx is None else class_idx              # gradcam.py:31-34
    score = logit[:, idx].squeeze()
    activations = acts["value"]
    (gradients,) = torch.autograd.grad(score, activations)
    b, k, u, v = gradients.shape
    alpha = gradients.view(b, k, -1).mean(2)                                # :39-41
    saliency = (alpha.view(b, k, 1, 1) * activations).sum(1, keepdim=True)  # :43
    saliency = F.relu(saliency)
    saliency = F.interpolate(saliency, size=x.shape[2:], mode="bilinear")   # :45
    lo, hi = saliency.min(), saliency.max()
    return ((saliency - lo) / (hi - lo)).detach()                           # :46-47


def gradcam_batch(model, x, class_idx=None):
    idx = (lambda i: None) if class_idx is None else (lambda i: int(class_idx[i]) if hasattr(class_idx, "__len__") else class_idx)
    return torch.cat([gradcam(model, x[i:i + 1], idx(i)) for i in range(x.shape[0])])
