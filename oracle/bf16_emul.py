"""Oracle: bf16-operand emulation of the precision=16 training path on torch-CPU.

The reference trains under pl.Trainer(precision=16) (src/self_supervised/tools.py:263): torch.autocast rounds the
operands of every conv / linear to 16 bits and accumulates in fp32.  The HIP path does the same with bf16 operands
(fp32 storage, rounding inside the kernels' loaders, forward AND backward).  This module restates that arithmetic
exactly -- fp32 math on bf16-rounded operands, including the gradients' operands -- so the kernels can be held to
a tight bound instead of the loose "tracks fp32" one.
"""
import torch
import torch.nn.functional as F
from torch import nn


def bf(t):
    return t.bfloat16().float()


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad):
        ctx.save_for_backward(x, w)
        ctx.sp = (stride, pad)
        return F.conv2d(bf(x), bf(w), None, stride, pad)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.sp
        xr, wr, dyr = bf(x).requires_grad_(), bf(w).requires_grad_(), bf(dy)
        with torch.enable_grad():
            y = F.conv2d(xr, wr, None, stride, pad)
        gx, gw = torch.autograd.grad(y, (xr, wr), dyr)
        return gx, gw, None, None


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return bf(x) @ bf(w).t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyr = bf(dy)
        return dyr @ bf(w), dyr.t() @ bf(x)


def emulate_bf16(model, fp32_stem=True):
    """Patches every Conv2d / Linear of ``model`` in place to compute on bf16-rounded operands (bias stays fp32).
    fp32_stem: the 3-channel conv1 stays fp32, as in the HIP path (its image-space kernels are fp32 MFMA in both
    precisions: 2 % of the step, and the raw pixels keep full precision)."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            if fp32_stem and m.in_channels == 3:
                continue
            m.forward = (lambda mod: lambda x: _Conv.apply(x, mod.weight, mod.stride[0], mod.padding[0]))(m)
        elif isinstance(m, nn.Linear):
            m.forward = (lambda mod: lambda x: _Linear.apply(x, mod.weight) + (mod.bias if mod.bias is not None else 0.0))(m)
    return model
