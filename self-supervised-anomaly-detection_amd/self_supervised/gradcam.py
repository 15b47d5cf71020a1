"""Grad-CAM localisation for the image-level branch -- same surface as src/self_supervised/gradcam.py of the reference
(`GradCam(model)(input, class_idx)` -> saliency [B][1][H][W] in [0, 1]).

The reference hooks layer4, back-propagates one class logit and averages the gradient over the 8x8 positions.
layer4's output reaches the logits only through the global average pool, so that gradient is the same at every
position: grad[b,k,u,v] = dpooled[b,k] / (U*V) and alpha = its spatial mean = dpooled[b,k] / (U*V).  Only the head's
input-gradient (a few [B x 512] MFMA GEMMs, no parameter gradients) is therefore needed; the contraction with the
activations, ReLU and the bilinear resize are two HIP launches (ssad_gradcam_map, ssad_blur_relu_bilinear ksize=1).

The reference only works with one image per call (`score.backward()` needs a scalar, gradcam.py:31-36) and normalises
with the min / max of the returned tensor; batches here give every image its own min / max, i.e. exactly the result of
calling the reference once per image as src/evaluator.py:270-279 does.
"""
import torch

from . import ops, training


class GradCam:
    def __init__(self, model):
        self.localizer = model
        self.localizer.eval()                                     # gradcam.py:10

    @torch.no_grad()
    def compute_gradcam(self, input_tensor, class_idx=None):
        m = self.localizer
        m.eval()
        x = input_tensor.to(next(m.parameters()).device, torch.float32).contiguous()
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected a [B][3][H][W] tensor")
        b, _, h, w = x.shape
        if h != w:
            raise NotImplementedError("square inputs only (the resize kernel writes target x target maps)")
        if getattr(m, "patch_level", False):
            raise RuntimeError("Grad-CAM belongs to the image-level branch: disable patch-level mode first")
        eng = training.get_engine(m)
        bf16, eng.bf16 = eng.bf16, False
        pg, eng.param_grads = eng.param_grads, False              # input gradients only: the fused conv + BatchNorm + ReLU epilogues, no kept z
        try:
            logits, _ = eng.forward(x)                            # eval statistics everywhere; keeps the head tape
            if class_idx is None:
                idx = logits.argmax(1)                            # gradcam.py:31-32
            else:
                idx = torch.as_tensor(class_idx, device=logits.device).long().reshape(-1).expand(b)
            dlogits = torch.zeros_like(logits)
            dlogits[torch.arange(b, device=logits.device), idx] = 1.0
            dpooled, act = eng.head_input_grad(dlogits)
        finally:
            eng.bf16, eng.param_grads = bf16, pg
        _, u, v, k = act.shape
        off = eng.gap_off["layer4"]
        alpha = dpooled[:, off:off + k] / float(u * v)            # spatial mean of the (constant) layer4 gradient, :39-41
        sal = ops.gradcam_map(act, alpha)                         # :43
        sal = ops.blur_relu_bilinear(sal, ksize=1, target=h)      # relu + F.interpolate(bilinear), :44-45
        lo = sal.amin(dim=(1, 2, 3), keepdim=True)
        hi = sal.amax(dim=(1, 2, 3), keepdim=True)
        return (sal - lo) / (hi - lo)                             # :46-47 (NaN for a constant map, as in the reference)

    def __call__(self, input, class_idx=None):
        return self.compute_gradcam(input, class_idx)
