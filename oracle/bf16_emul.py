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


class _ConvFwd16(torch.autograd.Function):
    """conv1 of the HIP path since round 4: the FORWARD runs on bf16-rounded operands (csrc/stem16.hip), its weight gradient on the
    fp32 image and fp32 dz (csrc/stem_wgrad.hip is the fp32-MFMA kernel in both precisions); the image needs no gradient."""
    @staticmethod
    def forward(ctx, x, w, stride, pad):
        ctx.save_for_backward(x, w)
        ctx.sp = (stride, pad)
        return F.conv2d(bf(x), bf(w), None, stride, pad)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.sp
        wr = w.detach().requires_grad_()
        with torch.enable_grad():
            y = F.conv2d(x.detach(), wr, None, stride, pad)
        gw, = torch.autograd.grad(y, wr, dy)
        return None, gw, None, None


def emulate_bf16(model, fp32_stem="wgrad"):
    """Patches every Conv2d / Linear of ``model`` in place to compute on bf16-rounded operands (bias stays fp32).
    fp32_stem: what the 3-channel conv1 does -- "wgrad" (the HIP path since round 4): forward on rounded operands, weight gradient in
    fp32; True (rounds 1-3): everything fp32; False: rounded like every other conv."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            if fp32_stem is True and m.in_channels == 3:
                continue
            if fp32_stem == "wgrad" and m.in_channels == 3:
                m.forward = (lambda mod: lambda x: _ConvFwd16.apply(x, mod.weight, mod.stride[0], mod.padding[0]))(m)
                continue
            m.forward = (lambda mod: lambda x: _Conv.apply(x, mod.weight, mod.stride[0], mod.padding[0]))(m)
        elif isinstance(m, nn.Linear):
            m.forward = (lambda mod: lambda x: _Linear.apply(x, mod.weight) + (mod.bias if mod.bias is not None else 0.0))(m)
    return model
