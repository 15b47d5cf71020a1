"""Training step on the HIP kernels: train-mode forward, hand-written backward, fused SGD, RCCL data parallel.

Restates ``PeraNet.training_step`` + ``configure_optimizers`` (src/self_supervised/models.py:256-277, :336-341
of the reference) and what ``pl.Trainer.fit`` does around them (autograd backward, optimizer step;
tools.py:260-303).  The reference is single-GPU; the data-parallel layer (one process per GPU, gradients
all-reduced with RCCL over xGMI in buckets that overlap the rest of backward) is new.

Memory layout: all parameters live in ONE flat fp32 arena ordered by backward completion (classifier first,
conv1 last); gradients and momentum are parallel arenas.  Conv weights are stored OHWI inside the arena and
exposed to PyTorch as OIHW-shaped (channels_last-strided) views, so ``state_dict()`` keeps the reference's
names and shapes while the kernels read them without repacking.  Bucketed all-reduce = contiguous arena
ranges; SGD = one launch per trainable range.
"""
import math
import os

import torch
import torch.distributed as dist
from torch import nn

from . import _hip, engine, ops

BN_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d)


# ---------------------------------------------------------------------------------------------
# parameter arena
# ---------------------------------------------------------------------------------------------
def _backward_order(model):
    """Parameters in the order their gradients become final during backward."""
    out = []
    out += [model.classifier.weight, model.classifier.bias]
    ls = list(model.latent_space)
    out += [ls[-1].weight, ls[-1].bias, ls[-2].weight, ls[-2].bias]
    for seq in reversed(ls[:-2]):
        out += [seq[1].weight, seq[1].bias, seq[0].weight]
    out += [model.concatenator[1].weight, model.concatenator[1].bias, model.concatenator[0].weight]
    head_count = len(out)
    fe = model.feature_extractor
    for name, _, _, _ in reversed(engine.BLOCKS):
        for blk in reversed(list(getattr(fe, name))):
            out += [blk.bn2.weight, blk.bn2.bias, blk.conv2.weight, blk.bn1.weight, blk.bn1.bias, blk.conv1.weight]
            if blk.downsample is not None:
                out += [blk.downsample[1].weight, blk.downsample[1].bias, blk.downsample[0].weight]
    out += [fe.bn1.weight, fe.bn1.bias, fe.conv1.weight]
    assert len(out) == len(list(model.parameters())), "parameter walk does not cover the model"
    return out, head_count


class ParamArena:
    def __init__(self, model):
        params, head_count = _backward_order(model)
        dev = params[0].device
        # every parameter starts at a multiple of ALIGN elements: 32-byte aligned floats, 16-byte aligned halves in the rounded copy
        # (`w16`: the half-tensor convs read their filters with 16-byte loads).  The pads stay zero in all three arenas (SGD maps
        # p = g = m = 0 to itself) and belong to the range of the parameter in front of them, so contiguous trainable parameters
        # still merge into one SGD launch / one all-reduce bucket.
        ALIGN = 8
        pad = lambda n: (n + ALIGN - 1) // ALIGN * ALIGN
        total = sum(pad(p.numel()) for p in params)
        self.p = torch.zeros(total, device=dev, dtype=torch.float32)
        self.g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.params, self.offset, self.span = params, {}, {}
        off = 0
        with torch.no_grad():
            for i, p in enumerate(params):
                n = p.numel()
                self.offset[id(p)] = (off, n)
                if p.dim() == 4:
                    o, c, kh, kw = p.shape
                    phys = self.p[off:off + n].view(o, kh, kw, c)
                    phys.copy_(p.detach().permute(0, 2, 3, 1))
                    p.data = phys.permute(0, 3, 1, 2)
                    p.grad = self.g[off:off + n].view(o, kh, kw, c).permute(0, 3, 1, 2)
                else:
                    flat = self.p[off:off + n].view(p.shape)
                    flat.copy_(p.detach())
                    p.data = flat
                    p.grad = self.g[off:off + n].view(p.shape)
                self.span[id(p)] = (off, off + pad(n))
                off += pad(n)
                if i + 1 == head_count:
                    self.head_end = off
        self.total = total

    def valid_for(self, model):
        p = self.params[0]
        return p.data_ptr() == self.p.data_ptr() and p.device == self.p.device

    def w(self, p):
        """Kernel view of a parameter: OHWI for 4-D weights, as-is otherwise."""
        off, n = self.offset[id(p)]
        if p.dim() == 4:
            o, c, kh, kw = p.shape
            return self.p[off:off + n].view(o, kh, kw, c)
        return self.p[off:off + n].view(p.shape)

    def grad(self, p):
        off, n = self.offset[id(p)]
        return self.g[off:off + n]

    def refresh_half(self):
        """Round the whole arena to halves once (start of a precision-16 step with half tensors): the copy its conv kernels read."""
        if getattr(self, "p16", None) is None:
            self.p16 = torch.empty(self.total, device=self.p.device, dtype=torch.float16)
        ops.cvt_f32_f16(self.p, self.p16)

    def w16(self, p):
        """Kernel view (OHWI) of a conv weight inside the half copy of the arena."""
        off, n = self.offset[id(p)]
        o, c, kh, kw = p.shape
        return self.p16[off:off + n].view(o, kh, kw, c)

    def sync_grads(self):
        """Compatibility path (PyTorch Lightning / set_to_none): make the arena hold whatever ``p.grad`` holds."""
        for p in self.params:
            off, n = self.offset[id(p)]
            if p.grad is None or not p.requires_grad:
                continue
            view = self.g[off:off + n]
            if p.grad.data_ptr() != view.data_ptr():
                src = p.grad.permute(0, 2, 3, 1) if p.dim() == 4 else p.grad
                view.view(src.shape).copy_(src)
                p.grad = view.view(src.shape).permute(0, 3, 1, 2) if p.dim() == 4 else view.view(p.shape)

    def trainable_ranges(self):
        """Merged [start, end) arena ranges whose parameters require grad (frozen ones are never touched)."""
        ranges = []
        for p in self.params:
            if not p.requires_grad:
                continue
            off, end = self.span[id(p)]          # the parameter and its alignment pad
            if ranges and ranges[-1][1] == off:
                ranges[-1][1] = end
            else:
                ranges.append([off, end])
        return [tuple(r) for r in ranges]


# ---------------------------------------------------------------------------------------------
# layers with saved state
# ---------------------------------------------------------------------------------------------
class _Affine:
    """conv/linear (+bias) followed by BatchNorm (train or eval statistics), optional residual and ReLU."""

    def __init__(self, eng, lin, bn, stride=1, pad=0, relu=False, stem=False):
        self.eng, self.lin, self.bn, self.stride, self.pad, self.relu, self.stem = eng, lin, bn, stride, pad, relu, stem
        self.is_conv = isinstance(lin, nn.Conv2d)
        self.eval_stats = False       # the last forward kept z under EVAL-mode statistics (trainable BatchNorm weight)

    def weight(self):
        a = self.eng.arena
        if self.stem:
            return ops.pack_stem_weight(a.w(self.lin.weight))        # straight from the arena's OHWI view (no OIHW copy per step)
        w = a.w(self.lin.weight)
        return w if self.is_conv else w.view(w.shape[0], 1, 1, w.shape[1])

    def _stem_fwd(self, img, w):
        """conv1 straight from the NCHW image (the unfused fp32-MFMA form; the training step takes fwd_pool); train or eval BatchNorm + ReLU."""
        bn = self.bn
        self.x, self.res_used = img, False
        self.eval_stats = False
        if bn.training:
            z = ops.stem_fwd(img, w, None, None, relu=False)
            mom = 0.1 if bn.momentum is None else bn.momentum
            self.mean, self.invstd = ops.bn_stats(z, 64, bn.eps, mom, bn.running_mean, bn.running_var)
            with torch.no_grad():
                self.eng.count_batch(bn)
            a = self.eng.arena
            y = ops.bn_apply_fwd(z, self.mean, self.invstd, a.w(bn.weight), a.w(bn.bias), None, True)
            self.z = z
        else:
            with torch.no_grad():
                self.invstd = torch.rsqrt(bn.running_var + bn.eps)
                self.mean = bn.running_mean
                scale = (bn.weight * self.invstd).contiguous()
                shift = (bn.bias - bn.running_mean * scale).contiguous()
            if self._gamma_grad_under_eval_stats():
                # eval-mode statistics but a TRAINABLE BatchNorm weight: its gradient needs the normalised input, so the conv output
                # is kept and the affine map applied by the BatchNorm kernel (running statistics in the role of mean / invstd)
                a = self.eng.arena
                z = ops.stem_fwd(img, w, None, None, relu=False)
                y = ops.bn_apply_fwd(z, self.mean, self.invstd.contiguous(), a.w(bn.weight), a.w(bn.bias), None, True)
                self.z, self.eval_stats = z, True
            else:
                y = ops.stem_fwd(img, w, scale, shift, relu=True)
                self.z = None
        self.y = y
        return y

    def _gamma_grad_under_eval_stats(self):
        bn = self.bn
        return (bn is not None and not bn.training and bn.weight.requires_grad and self.eng.param_grads and torch.is_grad_enabled())

    def fwd_pool(self, img):
        """Stem + max-pool: (pooled NHWC, argmax slots).  With batch statistics the BatchNorm + ReLU run inside the
        pooling kernel and the 128x128 activation is never stored (its backward recomputes the mask from z)."""
        bn = self.bn
        self.eval_stats = False       # a flag of the LAST forward only (set below where eval-mode statistics meet a trainable weight)
        if not bn.training:
            return ops.maxpool3x3s2_fwd_idx(self._stem_fwd(img, self.weight()))
        self.x, self.res_used = img, False
        mom = 0.1 if bn.momentum is None else bn.momentum
        if self.eng.bf16 in (1, 2) and self.eng.sw_stem16 and img.shape[2] >= 64 and img.shape[3] >= 64:
            # (smaller images are first resized to 64 x 64, models.py:217-219: the fp32 stem's loader does that, this kernel does not)
            # precision 16 / 'bf16': conv1 on the 16-bit matrix instructions (csrc/stem16.hip), as autocast runs it
            z, self.mean, self.invstd = ops.stem_fwd_stats16(img, self.eng.arena.w(self.lin.weight), bn.eps, mom, bn.running_mean,
                                                             bn.running_var, self.eng.bf16, out_half=self.eng.h16)
        else:
            z, self.mean, self.invstd = ops.stem_fwd_stats(img, self.weight(), bn.eps, mom, bn.running_mean, bn.running_var)
        with torch.no_grad():
            self.eng.count_batch(bn)
        a = self.eng.arena
        self.z, self.y = z, None
        self.zwin = None
        if self.eng.sw_poolwin and torch.is_grad_enabled() and self.eng.param_grads and self.eng.trunk_grad:
            # also the raw z of every window's winner: the backward pass takes the BatchNorm reduction over the pooled tensors
            pooled, idx, self.zwin = ops.bn_relu_maxpool_fwd(z, self.mean, self.invstd, a.w(bn.weight), a.w(bn.bias), winners=True)
            return pooled, idx
        return ops.bn_relu_maxpool_fwd(z, self.mean, self.invstd, a.w(bn.weight), a.w(bn.bias))

    def bwd_pool(self, idx, dpool, a0_shape):
        """Backward of fwd_pool: fills the BatchNorm and conv1 gradients (no input gradient: the image)."""
        a, bn = self.eng.arena, self.bn
        if self.z is None or getattr(self, "eval_stats", False):       # eval-mode statistics: unfused path
            self.bwd(ops.maxpool3x3s2_bwd_idx(idx, dpool, a0_shape), need_dx=False)
            return
        pg = self.eng.param_grads
        dbeta = a.grad(bn.bias) if (bn.bias.requires_grad and pg) else torch.empty(64, device=dpool.device)
        dgamma = a.grad(bn.weight) if (bn.weight.requires_grad and pg) else torch.empty(64, device=dpool.device)
        dz = ops.pool_bn_relu_bwd(idx, dpool.contiguous(), self.z, self.mean, self.invstd, a.w(bn.weight), a.w(bn.bias), dbeta, dgamma,
                                  zwin=getattr(self, "zwin", None))
        self.zwin = None
        if self.lin.weight.requires_grad and pg:
            with self.eng.wgrad_stream(dz, self.x):
                ops.stem_wgrad(self.x, dz, a.grad(self.lin.weight))
        self.x = self.z = self.y = None

    # ---- 64 -> 64 channel 3x3 / stride 1 layers (ResNet-18 layer1): halo-tile kernel, BatchNorm + ReLU of the first conv
    # of a block applied inside the second conv's input staging (csrc/conv_c64.hip) ----
    def c64_ok(self):
        l, bn = self.lin, self.bn
        return (self.is_conv and not self.stem and l.kernel_size == (3, 3) and self.stride == 1 and self.pad == 1
                and l.in_channels == 64 and l.out_channels == 64 and l.bias is None and bn is not None and bn.training
                and (not self.eng.bf16 or (self.eng.bf16 in (1, 2) and self.eng.sw_c64_16)) and self.eng.sw_c64)

    def fwd_raw_c64(self, x, producer=None):
        """conv + batch statistics only: returns the raw z of this layer and leaves its BatchNorm (+ ReLU) to the consumer.
        producer: the _Affine whose raw output `x` is -- its BatchNorm + ReLU is applied while x is staged (and, when the
        backward pass will need it, the normalised activation is emitted for the weight gradient)."""
        a, bn = self.eng.arena, self.bn
        mom = 0.1 if bn.momentum is None else bn.momentum
        st = (bn.eps, mom, bn.running_mean, bn.running_var)
        self.x_shape = tuple(x.shape)
        self.eval_stats = False
        n_, h_, w_, ci_ = x.shape
        if x.dtype == torch.float32 and self.eng.conv32w_ok(self, n_, h_, w_, ci_, self.lin.out_channels):
            wp = self.eng.packed_hw(self.lin, False, f32=True)         # launches that fill the chip: the register-fed form
            conv = lambda **kw: ops.conv3x3_hw(x, wp, self.lin.out_channels, **kw)
        else:
            conv = lambda **kw: ops.conv3x3_c64(x, a.w(self.lin.weight), bf16=self.eng.bf16, **kw)
        if producer is None:
            z, self.mean, self.invstd = conv(stats=st)
            self.x = x
        else:
            pb = producer.bn
            tr = (producer.mean, producer.invstd, a.w(pb.weight), a.w(pb.bias))
            need_x = self.lin.weight.requires_grad and self.eng.param_grads and torch.is_grad_enabled()
            if need_x:
                z, self.x, self.mean, self.invstd = conv(transform=tr, emit=True, stats=st)
            else:
                z, self.mean, self.invstd = conv(transform=tr, stats=st)
                self.x = None
        with torch.no_grad():
            self.eng.count_batch(bn)
        self.z, self.y, self.res_used, self.mask = z, None, False, None
        return z

    # ---- 3x3 / stride 1 / pad 1 layers of the precision-16 step with half tensors: halo-tile kernel of csrc/conv16.hip (persistent
    # workgroups, BatchNorm + ReLU of the producing layer applied while the input tile is staged) ----
    def conv16_ok(self):
        l, bn = self.lin, self.bn
        return (self.eng.h16 and self.eng.sw_conv16 and self.is_conv and not self.stem and l.kernel_size == (3, 3) and self.stride == 1
                and self.pad == 1 and l.bias is None and bn is not None and bn.training
                and bool(_hip.lib().ssad_conv3x3_h_ok(l.in_channels, l.out_channels)))

    def fwd_raw16(self, x, producer=None):
        """conv + batch statistics over half tensors: returns the raw z of this layer and leaves its BatchNorm (+ ReLU) to the
        consumer (apply_bn, or the next layer's input staging).  producer: the _Affine whose RAW output `x` is."""
        a, bn = self.eng.arena, self.bn
        mom = 0.1 if bn.momentum is None else bn.momentum
        st = (bn.eps, mom, bn.running_mean, bn.running_var)
        self.x_shape = tuple(x.shape)
        self.eval_stats = False
        n, h, wd, cin = x.shape
        cout = self.lin.out_channels
        if self.eng.sw_conv16w and ops.conv3x3_hw_ok(n, h, wd, cin, cout):      # launches that fill the chip: register-fed filters
            wp = self.eng.packed_hw(self.lin, False)
            conv = lambda **kw: ops.conv3x3_hw(x, wp, cout, **kw)
        else:
            w = a.w16(self.lin.weight)
            conv = lambda **kw: ops.conv3x3_h(x, w, **kw)
        if producer is None:
            z, self.mean, self.invstd = conv(stats=st)
            self.x = x
        else:
            pb = producer.bn
            tr = (producer.mean, producer.invstd, a.w(pb.weight), a.w(pb.bias))
            need_x = self.lin.weight.requires_grad and self.eng.param_grads and torch.is_grad_enabled()
            if need_x:
                z, self.x, self.mean, self.invstd = conv(transform=tr, emit=True, stats=st)
            else:
                z, self.mean, self.invstd = conv(transform=tr, stats=st)
                self.x = None
        with torch.no_grad():
            self.eng.count_batch(bn)
        self.z, self.y, self.res_used, self.mask = z, None, False, None
        return z

    def apply_bn(self, residual=None, use_mask=None):
        """The BatchNorm (+ residual) (+ ReLU) of a layer whose raw output fwd_raw_c64 left in self.z."""
        a, bn = self.eng.arena, self.bn
        if residual is not None and self.relu and (self.eng.use_relu_mask() if use_mask is None else use_mask):
            # the block's final ReLU leaves its active set as a nibble mask: backward never re-reads the activation
            y, self.mask = ops.bn_apply_fwd_mask(self.z, self.mean, self.invstd, a.w(bn.weight), a.w(bn.bias), residual, True)
            self.res_used, self.y = True, None
            return y
        y = ops.bn_apply_fwd(self.z, self.mean, self.invstd, a.w(bn.weight), a.w(bn.bias), residual, self.relu)
        self.res_used = residual is not None
        self.y = y if self.relu else None
        return y

    def fwd(self, x, residual=None, raw=False):
        """x NHWC (4-D; the stem takes the NCHW image).  Returns y NHWC.  raw (train-mode BatchNorm + ReLU, no residual): return the
        raw conv output z and leave the BatchNorm + ReLU to the consumer's input staging (fwd_raw16(producer=self))."""
        a, bn = self.eng.arena, self.bn
        w = self.weight()
        if self.stem:
            return self._stem_fwd(x, w)
        self.x, self.res_used = x, residual is not None
        self.x_shape = tuple(x.shape)
        self.mask = None
        self.eval_stats = False
        bias = getattr(self.lin, "bias", None)
        bf = self.eng.bf16
        if x.dtype == torch.float16:             # precision-16 step with half tensors: the rounded copy of the weights
            assert self.is_conv and bn is not None and bn.training and bias is None
            w = a.w16(self.lin.weight)
        if bn is None:
            y = ops.conv_fwd(x, w, None, a.w(bias) if bias is not None else None, None, self.relu, self.stride, self.pad, bf)
            self.z = self.y = None
            return y
        c = bn.num_features
        if bn.training:
            mom = 0.1 if bn.momentum is None else bn.momentum
            if not self.is_conv and residual is None and ops.bn_small_ok(x.numel() // x.shape[-1], c):
                # BatchNorm1d over a training batch's rows: Linear, then statistics + running statistics + apply in one launch
                z = ops.conv_fwd(x, w, None, a.w(bias) if bias is not None else None, None, False, self.stride, self.pad, bf)
                y, self.mean, self.invstd = ops.bn_small_fwd(z, a.w(bn.weight), a.w(bn.bias), bn.eps, mom, bn.running_mean,
                                                             bn.running_var, self.relu)
                with torch.no_grad():
                    self.eng.count_batch(bn)
                self.z, self.y = z, (y if self.relu else None)
                return y
            if bias is None and c % 4 == 0:
                # batch statistics taken in the conv epilogue (no second pass over z)
                if x.dtype == torch.float32 and x.dim() == 4 and self.eng.conv32w_ok(self, x.shape[0], x.shape[1], x.shape[2], x.shape[3], c):
                    z, self.mean, self.invstd = ops.conv3x3_hw(x, self.eng.packed_hw(self.lin, False, f32=True), c,
                                                               stats=(bn.eps, mom, bn.running_mean, bn.running_var))
                else:
                    z, self.mean, self.invstd = ops.conv_fwd_stats(x, w, bn.eps, mom, bn.running_mean, bn.running_var,
                                                                   self.stride, self.pad, bf)
            else:
                z = ops.conv_fwd(x, w, None, a.w(bias) if bias is not None else None, None, False, self.stride, self.pad, bf)
                self.mean, self.invstd = ops.bn_stats(z, c, bn.eps, mom, bn.running_mean, bn.running_var)
            with torch.no_grad():
                self.eng.count_batch(bn)
            self.z = z
            if raw:
                assert residual is None and self.relu
                self.y = None
                return z
            if residual is not None and self.relu and self.eng.use_relu_mask():
                y, self.mask = ops.bn_apply_fwd_mask(z, self.mean, self.invstd, a.w(bn.weight), a.w(bn.bias), residual, True)
                self.y = None
                return y
            y = ops.bn_apply_fwd(z, self.mean, self.invstd, a.w(bn.weight), a.w(bn.bias), residual, self.relu)
        else:
            with torch.no_grad():
                self.invstd = torch.rsqrt(bn.running_var + bn.eps)
                self.mean = bn.running_mean
                scale = (bn.weight * self.invstd).contiguous()
                shift = bn.bias - bn.running_mean * scale
                if bias is not None:
                    shift = shift + bias * scale
            if self._gamma_grad_under_eval_stats():
                # (as in _stem_fwd) frozen statistics, trainable affine parameters -- e.g. freeze_net(['backbone']) followed by
                # un-freezing the BatchNorm weights only: keep z, apply the affine map with the running statistics
                z = ops.conv_fwd(x, w, None, a.w(bias) if bias is not None else None, None, False, self.stride, self.pad, bf)
                y = ops.bn_apply_fwd(z, self.mean, self.invstd.contiguous(), a.w(bn.weight), a.w(bn.bias), residual, self.relu)
                self.z, self.eval_stats = z, True
                self.y = y if self.relu else None
                return y
            y = ops.conv_fwd(x, w, scale, shift.contiguous(), residual, self.relu, self.stride, self.pad, bf)
            self.z = None
        self.y = y if self.relu else None
        return y

    def bwd(self, dy, need_dx=True, dx_residual=None, want_dres=False, dy_mask=None, dx_res_mask=None):
        """dy NHWC grad of the output.  Returns (dx or None, dres or None).
        dres: the gradient of the identity branch -- a tensor, or the pair (dy, nibble mask) when the forward left the final
        ReLU's active set as a mask (the consumer applies it: `dy_mask` of a downsample layer's bwd, `dx_res_mask` of the
        first conv's dgrad); dy_mask: this layer's dy is to be taken under that mask."""
        a, bn = self.eng.arena, self.bn
        dres = None
        bias = getattr(self.lin, "bias", None)
        mask = getattr(self, "mask", None)
        if bn is None:
            dz = dy
        else:
            c = bn.num_features
            train_stats = self.z is not None and not getattr(self, "eval_stats", False)
            pg = self.eng.param_grads
            wg, bg = bn.weight.requires_grad and pg, bn.bias.requires_grad and pg
            small_bias = None
            if (train_stats and not self.is_conv and dy_mask is None and mask is None and not want_dres and not self.res_used
                    and ops.bn_small_ok(dy.numel() // c, c)):
                # the one-launch form of the two branches below that a BatchNorm1d of the head can take (ReLU without residual:
                # mask from z; or no ReLU), with the bias gradient of the Linear in front (column sums of dz) in the same launch
                dbeta = a.grad(bn.bias) if bg else None
                dgamma = a.grad(bn.weight) if wg else None
                small_bias = a.grad(bias) if (bias is not None and bias.requires_grad and pg) else None
                dz = ops.bn_small_bwd(dy, self.z, self.mean, self.invstd, a.w(bn.weight), a.w(bn.bias) if self.relu else None,
                                      dbeta, dgamma, small_bias)
            elif train_stats and (dy_mask is not None or (mask is not None and want_dres)):
                # residual block without re-reading activations: g = dy * mask inside the two BatchNorm passes
                dbeta = a.grad(bn.bias) if bg else torch.empty(c, device=dy.device)
                dgamma = a.grad(bn.weight) if wg else torch.empty(c, device=dy.device)
                mk = dy_mask if dy_mask is not None else mask
                dz = ops.bn_bwd_mask(dy, mk, self.z, self.mean, self.invstd, a.w(bn.weight), dbeta, dgamma)
                if want_dres:
                    dres = (dy, mk)
            elif train_stats and self.relu and not self.res_used and not want_dres:
                # y = relu(bn(z)) with nothing added in between: take the mask from z, skip re-reading y
                dbeta = a.grad(bn.bias) if bg else torch.empty(c, device=dy.device)
                dgamma = a.grad(bn.weight) if wg else torch.empty(c, device=dy.device)
                dz = ops.bn_bwd_zmask(dy, self.z, self.mean, self.invstd, a.w(bn.weight), a.w(bn.bias), dbeta, dgamma)
            elif train_stats or wg or bg:
                dbeta = a.grad(bn.bias) if bg else torch.empty(c, device=dy.device)
                dgamma = a.grad(bn.weight) if wg else torch.empty(c, device=dy.device)
                zsrc = self.z             # train-mode statistics, or eval-mode statistics with z kept for a trainable weight
                assert zsrc is not None or not wg, "the forward keeps z whenever the BatchNorm weight is trainable"
                ops.bn_bwd_reduce(dy, self.y, zsrc, self.mean, self.invstd, dbeta, dgamma, c)
                dz, dres = ops.bn_apply_bwd(dy, self.y, self.z, self.mean, self.invstd, a.w(bn.weight), dbeta, dgamma,
                                            want_dres, eval_mode=not train_stats)
            else:
                dz, dres = ops.bn_apply_bwd(dy, self.y, self.z, self.mean, self.invstd, a.w(bn.weight), None, None,
                                            want_dres, eval_mode=True)
        cout = dz.shape[-1]
        if bias is not None and bias.requires_grad and self.eng.param_grads and not (bn is not None and small_bias is not None):
            ops.bn_bwd_reduce(dz, None, None, None, None, a.grad(bias), None, cout)
        bf = self.eng.bf16
        if self.lin.weight.requires_grad and self.eng.param_grads:
            with self.eng.wgrad_stream(dz, self.x):       # weight gradients run beside the input-gradient chain
                if self.stem:
                    ops.stem_wgrad(self.x, dz, a.grad(self.lin.weight))
                elif self.is_conv:
                    k = self.lin.kernel_size[0]
                    ops.conv_wgrad(dz, self.x, a.grad(self.lin.weight), k, k, self.stride, self.pad, bf16=bf)
                else:
                    ops.conv_wgrad(dz, self.x, a.grad(self.lin.weight), 1, 1, 1, 0, bf16=bf)
        dx = None
        if need_dx:
            w = self.weight()
            dzz, wt = dz, w
            small = (int(bf) in (0, 1, 2) and not self.is_conv and cout % 4 == 0 and w.data_ptr() % 16 == 0 and
                     dz.numel() // cout <= _hip.lib().ssad_linear_small_max_rows())
            if cout % 32 and not small:         # classifier: pad the contraction to a multiple of 32 with zeros (the small-batch
                                                # linear kernel takes any multiple of 4)
                padc = (cout + 31) // 32 * 32
                dzz = torch.zeros(dz.shape[:-1] + (padc,), device=dz.device)
                dzz[..., :cout] = dz
                wt = torch.zeros((padc,) + tuple(w.shape[1:]), device=dz.device)
                wt[:cout] = w
            half = dz.dtype == torch.float16
            if half and dz.dim() == 4 and self.conv16_ok():
                n_, h_, w_, _ = dzz.shape
                if self.eng.sw_conv16w and ops.conv3x3_hw_ok(n_, h_, w_, cout, self.lin.in_channels):
                    dx = ops.conv3x3_hw(dzz, self.eng.packed_hw(self.lin, True), self.lin.in_channels, residual=dx_residual,
                                        res_mask=dx_res_mask)
                else:
                    assert dx_res_mask is None, "use_relu_mask16 keeps the mask away from launches csrc/conv16.hip runs"
                    dx = ops.conv3x3_h(dzz, self.eng.flipped(self.lin, wt, half=True), residual=dx_residual)
            elif (dz.dtype == torch.float32 and dz.dim() == 4 and wt is w and self.is_conv
                  and self.eng.conv32w_ok(self, dz.shape[0], dz.shape[1], dz.shape[2], cout, self.lin.in_channels)):
                dx = ops.conv3x3_hw(dzz.contiguous(), self.eng.packed_hw(self.lin, True, f32=True), self.lin.in_channels,
                                    residual=dx_residual, res_mask=dx_res_mask)
            elif self.c64_ok() and dz.dim() == 4:
                dx = ops.conv3x3_c64(dzz, self.eng.flipped(self.lin, wt), residual=dx_residual, res_mask=dx_res_mask, bf16=bf)
            elif half:
                assert dx_res_mask is None
                dx = ops.conv_dgrad(dzz, self.eng.flipped(self.lin, wt, half=True), self.x_shape, self.stride, self.pad, dx_residual, bf)
            else:
                wf = self.eng.flipped(self.lin, wt) if wt is w else ops.flip_transpose_weight(wt)
                dx = ops.conv_dgrad(dzz, wf, self.x_shape, self.stride, self.pad, dx_residual, bf, res_mask=dx_res_mask)
        self.x = self.z = self.y = self.mask = None
        return dx, dres


class _Aside:
    """Second HIP stream for the launches of a step that nothing on its critical path waits for (ops.ASIDE): each launch is a fork
    -- the side stream waits for the main stream's position, then runs the kernel -- and `join` makes the main stream wait for
    everything forked so far.  Recorded into the step's hipGraph the forks become parallel branches: a 5-12 us slab reduction or
    pooling kernel runs beside the next convolution instead of in front of it.  Operands are kept alive until the join (a block
    freed earlier could be handed to a main-stream kernel that runs concurrently with the side kernel still reading it)."""

    def __init__(self):
        self.stream, self.keep, self.dirty = None, [], False

    def launch(self, fn, keep=()):
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            fn()
        self.keep.extend(t for t in keep if t is not None)
        self.dirty = True

    def join(self):
        if self.dirty:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.dirty = False
        self.keep = []


class TrainEngine:
    """Owns the arenas and the per-step tape for one PeraNet."""

    def __init__(self, model):
        self.model = model
        self.arena = ParamArena(model)
        # precision=16 (the reference's pl.Trainer setting, tools.py:263): conv / linear operands are rounded to bf16 in
        # the kernels' loaders (fp32 tensors, master weights, accumulation, BatchNorm, loss and SGD stay fp32)
        self.bf16 = False
        self.param_grads = True       # False: backward produces input gradients only (Grad-CAM)
        fe = model.feature_extractor
        self.stem = _Affine(self, fe.conv1, fe.bn1, 2, 3, True, stem=True)   # dedicated image-space kernels
        self.blocks = []
        for name, _, _, _ in engine.BLOCKS:
            for blk in getattr(fe, name):
                d = {"name": name,
                     "c1": _Affine(self, blk.conv1, blk.bn1, blk.stride, 1, True),
                     "c2": _Affine(self, blk.conv2, blk.bn2, 1, 1, True),
                     "ds": None}
                if blk.downsample is not None:
                    d["ds"] = _Affine(self, blk.downsample[0], blk.downsample[1], blk.stride, 0, False)
                self.blocks.append(d)
        self.head = [_Affine(self, model.concatenator[0], model.concatenator[1])]
        ls = list(model.latent_space)
        for seq in ls[:-2]:
            self.head.append(_Affine(self, seq[0], seq[1], relu=True))
        self.head.append(_Affine(self, ls[-2], ls[-1]))
        self.cls = _Affine(self, model.classifier, None)
        self.gap_off, off = {}, 0
        for k in ("layer1", "layer2", "layer3"):
            if k in model.layer_outputs:
                self.gap_off[k] = off
                off += {"layer1": 64, "layer2": 128, "layer3": 256}[k]
        self.gap_off["layer4"] = off
        self.pooled_dim = off + 512
        self.bucket_hooks = None      # set by DataParallelStep: callable(range_end) after each finished arena prefix
        # Weight gradients only feed the optimiser (and the gradient all-reduce), the input-gradient chain never waits for
        # them, so they CAN be launched on a second HIP stream (SSAD_WGRAD_STREAM=1) to fill the tails of the dgrad /
        # BatchNorm kernels.  Measured (round 2, bs256 / bs32): 36.33 vs 36.07 ms and 8.05 vs 7.51 ms per step -- every big
        # kernel already fills the chip, the second stream only adds dependencies -- so it is OFF by default.
        self._nbt = []
        self._flip_ready = False
        self.side = None
        self._side_keep = []
        # environment switches are read ONCE here: a captured step (hipGraph plan) must never be replayed under a different
        # setting than the one it was recorded with (they are part of the plan key as well)
        self._side_on = os.environ.get("SSAD_WGRAD_STREAM", "0") == "1"
        self.sw_c64 = os.environ.get("SSAD_C64", "1") != "0"
        self.sw_relu_mask = os.environ.get("SSAD_RELU_MASK", "1") != "0"
        self.sw_wgrad_halo = os.environ.get("SSAD_WGRAD_HALO", "1") != "0"
        self.sw_stem16 = os.environ.get("SSAD_STEM16", "1") != "0"
        self.sw_c64_16 = os.environ.get("SSAD_C64_16", "1") != "0"
        # precision 16: the trunk's activations (and their gradients) live in HBM as halves, as torch.autocast stores them
        # (SSAD_ACT16=0: fp32 tensors with operands rounded while staged -- rounds 2-4)
        self.sw_act16 = os.environ.get("SSAD_ACT16", "1") != "0"
        self.sw_conv16 = os.environ.get("SSAD_CONV16", "1") != "0"
        self.sw_conv16w = os.environ.get("SSAD_CONV16W", "1") != "0"     # register-fed form of the same conv (csrc/conv16w.hip)
        self.sw_conv32w = os.environ.get("SSAD_CONV32W", "1") != "0"     # ... and its exact-fp32 instantiation (the fp32 step)
        self.sw_raw32 = os.environ.get("SSAD_RAW32", "1") != "0"         # fp32: bn1 + ReLU inside conv2's staging on every block it runs
        self.sw_poolwin = os.environ.get("SSAD_POOLWIN", "1") != "0"     # stem: BatchNorm backward reduction over the pooled tensors
        # off-critical-path launches (slab reductions, head weight gradients, pooling rows, the filter tables of the backward pass)
        # as parallel branches of the step (ops.ASIDE; DataParallelStep switches it on for its steps)
        # the slab reductions of a step's weight gradients in ONE launch at the end of backward (ssad_wgrad_reduce_batch).  OFF: measured
        # (round 6, same box, two alternating runs) fp32 32.34 -> 32.38 ms, batch 32 5.424 -> 5.432, precision 16 8.76 -> 8.80: the 19 per-layer
        # reductions are not launch-bound but byte-bound (40 MB each at ~5 TB/s, read while the slabs are still in the Infinity Cache;
        # deferred, they come from HBM).  Kept as a switch: it takes 18 dispatches off the step at equal time.
        self.sw_batch_reduce = os.environ.get("SSAD_BATCH_REDUCE", "0") != "0"
        self.sw_mask16 = os.environ.get("SSAD_MASK16", "1") != "0"       # precision 16, half tensors: nibble masks for the residual blocks
        self.sw_aside = int(os.environ.get("SSAD_ASIDE", "0"))        # 0 off, 1 every such launch, 2 only the filter tables beside the stem
        self.aside = _Aside()
        self._tables_used = set()     # filter tables the last backward asked for: requested ahead, beside the stem, by the next forward
        self._packed = {}             # packed 3x3 filters, key (flip, f32): forward / input-gradient (flipped) tables, halves / floats
        self._packed_ready = {}
        self.h16 = False              # decided per forward (trunk BatchNorms in training mode, whole images of >= 64 x 64)
        self._flip16_view, self._flip16_ready = {}, False

    def switches(self):
        """Everything besides shapes that decides which launches a step consists of (hipGraph plan key)."""
        return (self.bf16, self.param_grads, self.sw_c64, self.sw_relu_mask, self.sw_wgrad_halo, self.sw_stem16, self.sw_c64_16, self.sw_act16, self.sw_conv16, self.sw_conv16w, self.sw_conv32w, self.sw_raw32, self.sw_poolwin, self._side_on, self.sw_aside, self.sw_mask16, self.sw_batch_reduce,
                os.environ.get("SSAD_WGRAD_HALO", "1") != "0", torch.is_grad_enabled())

    # ---- second stream for the weight gradients ----
    class _Side:
        def __init__(self, eng, keep):
            self.eng, self.keep, self.ctx = eng, keep, None

        def __enter__(self):
            eng = self.eng
            if not eng._side_on or ops.PROFILE is not None:
                return self
            if eng.side is None:
                eng.side = torch.cuda.Stream()
            eng._side_keep.extend(t for t in self.keep if t is not None)   # operands stay allocated until the join
            eng.side.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(eng.side)
            self.ctx.__enter__()
            return self

        def __exit__(self, *exc):
            if self.ctx is not None:
                self.ctx.__exit__(*exc)
            return False

    def count_batch(self, bn):
        """num_batches_tracked += 1 for every train-mode BatchNorm of the step, in ONE multi-tensor launch (end of forward)."""
        self._nbt.append(bn.num_batches_tracked)

    def _flipped_half(self, lin):
        """The same table over the block convs only, written as halves (precision-16 step with half tensors)."""
        key = id(lin.weight)
        if self._flip16_ready:
            return self._flip16_view[key]
        if not self._flip16_view:
            import ctypes
            a, desc, off, shapes = self.arena, [], 0, {}
            for layer in [d[k] for d in self.blocks for k in ("c1", "c2", "ds") if d[k] is not None]:
                p = layer.lin.weight
                o, c, kh, kw = p.shape
                desc += [a.offset[id(p)][0], off, o, c, kh, kw]
                shapes[id(p)] = (off, (c, kh, kw, o))
                off += p.numel()
            self._flip16_buf = torch.empty(off, device=a.p.device, dtype=torch.float16)
            self._flip16_desc = (ctypes.c_int64 * len(desc))(*desc)
            self._flip16_n = len(desc) // 6
            self._flip16_view = {k: self._flip16_buf[o:o + s[0] * s[1] * s[2] * s[3]].view(s) for k, (o, s) in shapes.items()}
        _hip.check(_hip.lib().ssad_flip_transpose_batch_h(_hip.ptr(self.arena.p), self._flip16_buf.data_ptr(), self._flip16_desc,
                                                          self._flip16_n, _hip.stream()))
        self._flip16_ready = True
        self._tables_used.add(("flip16",))
        return self._flip16_view[key]

    def packed_hw(self, lin, flip, f32=False):
        """The 3x3 filter of a block conv in the fragment order ssad_conv3x3_hw / _fw reads (flip: as the input gradient's filter; f32: the
        pack of the exact-fp32 form).  All block convs are packed by ONE launch per table and step, from the fp32 master weights (the
        first request after a forward)."""
        key = (bool(flip), bool(f32))
        tab = self._packed.get(key)
        if tab is None:
            entries, keys = [], []
            for layer in [d[k] for d in self.blocks for k in ("c1", "c2")]:
                p = layer.lin.weight
                o, c, kh, kw = p.shape
                if (kh, kw) != (3, 3) or layer.stride != 1 or o % 64 or c % 64:
                    continue
                entries.append((self.arena.offset[id(p)][0], c, o, True) if flip else (self.arena.offset[id(p)][0], o, c, False))
                keys.append(id(p))
            sizes = [e[1] * 9 * e[2] for e in entries]
            buf = torch.empty(sum(sizes), device=self.arena.p.device, dtype=torch.float32 if f32 else torch.float16)
            offs = [sum(sizes[:i]) for i in range(len(sizes))]
            tab = self._packed[key] = {"buf": buf, "entries": entries,
                                       "view": {k: buf[o:o + n] for k, o, n in zip(keys, offs, sizes)}}
        if not self._packed_ready.get(key, False):
            ops.conv3x3_hw_pack(self.arena.p, tab["entries"], out=tab["buf"], f32=f32)
            self._packed_ready[key] = True
            self._tables_used.add(("pack",) + key)
        return tab["view"][id(lin.weight)]

    def conv32w_ok(self, layer, n, h, w, cin, cout):
        """Whether the exact-fp32 step runs this 3x3 / stride 1 conv (forward: cin -> cout; input gradient: the roles swapped) on the
        register-fed form (csrc/conv16w.hip, T = float)."""
        l = layer.lin
        return (self.sw_conv32w and not self.bf16 and layer.is_conv and not layer.stem and l.kernel_size == (3, 3) and layer.stride == 1
                and layer.pad == 1 and l.bias is None and ops.conv3x3_hw_ok(n, h, w, cin, cout, f32=True))

    def flipped(self, lin, w, half=False):
        """dgrad operand of a conv layer (ssad_flip_transpose_weight of its OHWI weight).  All block convs are flipped by one
        launch per step (the first request after a forward), not one launch per layer."""
        if half:
            return self._flipped_half(lin)
        key = id(lin.weight)
        # the fp32 table holds every layer, or -- while the trunk runs on half tensors and takes its flipped filters from the half
        # table -- the head only.  One table per mode, BOTH kept alive: a recorded step (hipGraph plan) holds the addresses of the
        # table it was captured with
        tabs = self.__dict__.setdefault("_flip_tables", {})
        tab = tabs.get(self.h16)
        if tab is None:                              # build the table once: arena offsets of every layer it covers
            import ctypes
            a, desc, off, shapes = self.arena, [], 0, {}
            trunk = [] if self.h16 else [d[k] for d in self.blocks for k in ("c1", "c2", "ds") if d[k] is not None]
            for layer in trunk + list(self.head) + [self.cls]:      # block convs; the head's and classifier's linear layers as 1 x 1 filters
                p = layer.lin.weight
                o, c, kh, kw = p.shape if p.dim() == 4 else (p.shape[0], p.shape[1], 1, 1)
                desc += [a.offset[id(p)][0], off, o, c, kh, kw]
                shapes[id(p)] = (off, (c, kh, kw, o))
                off += p.numel()
            buf = torch.empty(off, device=a.p.device, dtype=torch.float32)
            tab = tabs[self.h16] = {"buf": buf, "desc": (ctypes.c_int64 * len(desc))(*desc), "n": len(desc) // 6,
                                    "view": {k: buf[o:o + sh[0] * sh[1] * sh[2] * sh[3]].view(sh) for k, (o, sh) in shapes.items()}}
        if key not in tab["view"]:
            return ops.flip_transpose_weight(w)
        if self._flip_ready is not tab:              # first request after a forward: one launch for the whole table
            _hip.check(_hip.lib().ssad_flip_transpose_batch(_hip.ptr(self.arena.p), _hip.ptr(tab["buf"]), tab["desc"], tab["n"], _hip.stream()))
            self._flip_ready = tab
            self._tables_used.add(("flip32",))
        return tab["view"][key]

    def use_relu_mask(self):
        """Residual blocks keep their final ReLU's active set as a nibble mask (exact fp32 path, gradients wanted)."""
        return not self.bf16 and self.trunk_grad and torch.is_grad_enabled() and self.sw_relu_mask

    def use_relu_mask16(self, block_in, has_ds):
        """The same for a block of the precision-16 step with half tensors (round 6): whoever consumes the identity-branch gradient
        (dy, mask) must apply the mask itself -- the downsample layer's BatchNorm kernels do, and so does the register-fed conv
        (csrc/conv16w.hip) as the first conv's input gradient; launches too small for it (csrc/conv16.hip) keep the activation."""
        if not (self.h16 and self.sw_mask16 and self.trunk_grad and torch.is_grad_enabled() and self.sw_relu_mask):
            return False
        n, h, w, c = block_in.shape
        return bool(has_ds or (self.sw_conv16w and ops.conv3x3_hw_ok(n, h, w, c, c)))

    def wgrad_stream(self, *operands):
        return TrainEngine._Side(self, operands)

    def join_wgrad(self):
        """The compute stream waits for every weight-gradient launch issued so far (before an all-reduce of those ranges /
        before the optimiser); the operands kept alive for the side stream are released."""
        if self.side is not None and self._side_keep:
            torch.cuda.current_stream().wait_stream(self.side)
        self._side_keep = []
        ops.flush_reductions()        # the slab reductions collected since the last join: one launch
        self.aside.join()

    # ---- forward (models.py:210-253, train mode) ----
    def forward(self, x):
        m = self.model
        b, _, h, w = x.shape
        self.trunk_grad = any(p.requires_grad for p in m.feature_extractor.parameters())
        self._nbt, self._flip_ready, self._flip16_ready = [], False, False
        self._packed_ready = {}
        self.h16 = bool(self.bf16 == 2 and self.sw_act16 and self.sw_stem16 and h >= 64 and w >= 64 and self.param_grads and
                        all(mod.training for mod in m.feature_extractor.modules() if isinstance(mod, BN_TYPES)))
        aside = ops.ASIDE if (ops.ASIDE is not None and ops.PROFILE is None) else None
        if self.h16:
            if aside is not None:
                aside.launch(self.arena.refresh_half)
            else:
                self.arena.refresh_half()
        if aside is not None and self.trunk_grad and self.param_grads and torch.is_grad_enabled():
            # the filter tables the LAST step's passes asked for (flipped filters of the input gradients, fragment-order packs) only
            # depend on the parameters: built now, beside the stem, instead of in front of the first kernel that reads them
            c0 = self.blocks[0]["c1"].lin
            for t in sorted(self._tables_used):
                if t == ("flip32",):
                    aside.launch(lambda: self.flipped(self.cls.lin, None))
                elif t == ("flip16",) and self.h16:
                    aside.launch(lambda: self._flipped_half(c0))
                elif t[0] == "pack" and (t[2] or self.h16):
                    aside.launch(lambda t=t: self.packed_hw(c0, t[1], f32=t[2]))
        a, self.pool_idx = self.stem.fwd_pool(x.contiguous())
        if aside is not None:
            aside.join()                      # layer1 reads the packs / the rounded weights
            if self.sw_aside == 2:            # only the filter tables run beside the stem: everything else stays on the chain
                ops.ASIDE = None
        if not self.trunk_grad:
            self.stem.x = None
        _, _, _, ho, wo = ops.stem_geometry(h, w, 0, 0)
        self.a0_shape = (b, ho, wo, 64)
        pooled = torch.empty((b, self.pooled_dim), device=x.device, dtype=torch.float32)
        self.stage_shapes = {}
        for i, d in enumerate(self.blocks):
            idt = a
            if d["ds"] is not None:
                idt = d["ds"].fwd(a)
            if self.h16 and d["c2"].conv16_ok():
                # half tensors: conv2 (and conv1 where it has stride 1) on the halo-tile kernel; bn1 + ReLU inside conv2's input staging
                z1 = d["c1"].fwd_raw16(a) if d["c1"].conv16_ok() else d["c1"].fwd(a, raw=True)
                d["c2"].fwd_raw16(z1, producer=d["c1"])
                a = d["c2"].apply_bn(residual=idt, use_mask=self.use_relu_mask16(a, d["ds"] is not None))
            elif d["ds"] is None and d["c1"].c64_ok() and d["c2"].c64_ok():
                z1 = d["c1"].fwd_raw_c64(a)                       # bn1 + ReLU happen inside conv2's input staging
                d["c2"].fwd_raw_c64(z1, producer=d["c1"])
                a = d["c2"].apply_bn(residual=idt)
            elif (self.sw_raw32 and not self.bf16 and a.dtype == torch.float32 and d["c1"].bn.training and d["c2"].bn.training and
                  self.conv32w_ok(d["c2"], a.shape[0], (a.shape[1] - 1) // d["c1"].stride + 1, (a.shape[2] - 1) // d["c1"].stride + 1,
                                  d["c2"].lin.in_channels, d["c2"].lin.out_channels)):
                # exact fp32, launches that fill the chip: the same hand-over on every block -- conv1 leaves its raw output and its batch
                # statistics, conv2 (register-fed form) applies bn1 + ReLU while it stages its input and emits the activation its weight
                # gradient reads: one pass over the activation less per block than a separate BatchNorm apply
                z1 = d["c1"].fwd(a, raw=True)
                d["c2"].fwd_raw_c64(z1, producer=d["c1"])
                a = d["c2"].apply_bn(residual=idt)
            else:
                t = d["c1"].fwd(a)
                a = d["c2"].fwd(t, residual=idt)
            if i % 2 == 1 and d["name"] in self.gap_off:
                ops.gap_fwd(a, pooled, self.gap_off[d["name"]])
                self.stage_shapes[d["name"]] = a.shape
        self.last_shape = a.shape
        self.last_act = a             # layer4 output (NHWC): Grad-CAM's activations
        if not self.trunk_grad:
            self._drop_trunk_tape()
        self.aside.join()                     # the pooled rows (ops.gap_fwd may have run beside the convs)
        f = pooled.view(b, 1, 1, -1)
        for layer in self.head:
            f = layer.fwd(f)
        logits = self.cls.fwd(f)
        self.batch = b
        if self._nbt:
            with torch.no_grad():
                torch._foreach_add_(self._nbt, 1)
            self._nbt = []
        return logits.view(b, -1), f.view(b, -1)

    # ---- backward ----
    def backward(self, dlogits):
        """dlogits [B][num_classes] -> fills the gradient arena."""
        b = self.batch
        hook = self.bucket_hooks

        def notify(end):
            if hook is not None:
                self.join_wgrad()          # the bucket that may go out now must hold finished weight gradients
                hook(end)
        a = self.arena
        head_dx = self.trunk_grad or any(p.requires_grad for l in self.head for p in l.lin.parameters())
        d, _ = self.cls.bwd(dlogits.view(b, 1, 1, -1), need_dx=head_dx)
        for i, layer in enumerate(reversed(self.head)):
            last = i == len(self.head) - 1
            if d is None:
                break
            d, _ = layer.bwd(d, need_dx=(self.trunk_grad if last else True))
        notify(a.head_end)
        if not self.trunk_grad:
            self.join_wgrad()
            self._drop_tape()
            return
        dpooled = d.view(b, -1).contiguous()
        dy = torch.empty(self.last_shape, device=dpooled.device, dtype=torch.float16 if self.h16 else torch.float32)
        ops.gap_bwd(dpooled, dy, self.gap_off["layer4"], accumulate=False)
        for i in range(len(self.blocks) - 1, -1, -1):
            blk = self.blocks[i]
            dz2_dx, dres = blk["c2"].bwd(dy, need_dx=True, want_dres=True)
            masked = isinstance(dres, tuple)           # (dy, nibble mask): the identity-branch gradient, not materialised
            rmask = None
            if blk["ds"] is not None:
                dx_id, _ = blk["ds"].bwd(dres[0] if masked else dres, need_dx=True, dy_mask=dres[1] if masked else None)
            elif masked:
                dx_id, rmask = dres
            else:
                dx_id = dres
            dy, _ = blk["c1"].bwd(dz2_dx, need_dx=True, dx_residual=dx_id, dx_res_mask=rmask)
            if i % 2 == 0 and i > 0:
                prev = self.blocks[i - 1]["name"]
                if prev in self.gap_off:
                    ops.gap_bwd(dpooled, dy, self.gap_off[prev], accumulate=True)
            if i % 2 == 0:
                notify(a.span[id(blk["c1"].lin.weight if blk["ds"] is None else blk["ds"].lin.weight)][1])
        self.stem.bwd_pool(self.pool_idx, dy, self.a0_shape)
        notify(a.total)
        self.join_wgrad()
        self._drop_tape()

    def head_input_grad(self, dlogits):
        """d(sum(logits * dlogits)) / d(pooled features) [B][pooled_dim]: the head's backward without touching any
        parameter gradient (gradcam.py:36: score.backward() up to the layer4 hook).  Consumes the tape."""
        b = self.batch
        pg, self.param_grads = self.param_grads, False
        try:
            d, _ = self.cls.bwd(dlogits.view(b, 1, 1, -1), need_dx=True)
            for layer in reversed(self.head):
                d, _ = layer.bwd(d, need_dx=True)
        finally:
            self.param_grads = pg
        act = self.last_act
        self._drop_tape()
        self.last_act = None
        return d.view(b, -1), act

    def _drop_trunk_tape(self):
        self.pool_idx = None
        for d in self.blocks:
            for k in ("c1", "c2", "ds"):
                if d[k] is not None:
                    d[k].x = d[k].z = d[k].y = d[k].mask = None
                    d[k].eval_stats = False
        self.stem.x = self.stem.z = self.stem.y = None
        self.stem.eval_stats = False

    def _drop_tape(self):
        self._drop_trunk_tape()
        for l in self.head + [self.cls]:
            l.x = l.z = l.y = None
            l.eval_stats = False


def get_engine(model):
    eng = getattr(model, "_train_engine", None)
    if eng is None or not eng.arena.valid_for(model):
        eng = TrainEngine(model)
        model._train_engine = eng
        model._plan = None
    return eng


# ---------------------------------------------------------------------------------------------
# module-level API used by models.PeraNet
# ---------------------------------------------------------------------------------------------
def forward_train(model, x):
    eng = get_engine(model)
    logits, emb = eng.forward(x)
    return {"classifier": logits, "latent_space": emb}


def cross_entropy_eval(logits, y):
    la = ops.softmax_ce(logits.contiguous(), y.to(logits.device).contiguous())
    return la[0], la[1]


class _HipLoss(torch.autograd.Function):
    """Lets ``loss.backward()`` (PyTorch Lightning, plain torch optimisers) drive the HIP backward pass:
    the parameters are declared as inputs, their gradients are read back from the arena."""

    @staticmethod
    def forward(ctx, model, loss, dlogits, *params):
        ctx.model, ctx.dlogits, ctx.params = model, dlogits, params
        return loss.clone()

    @staticmethod
    def backward(ctx, gout):
        eng = get_engine(ctx.model)
        eng.backward(ctx.dlogits * gout)
        grads = tuple(p.grad.clone() if p.requires_grad else None for p in ctx.params)
        eng.arena.g.zero_()          # autograd now accumulates the returned gradients into p.grad (= arena views)
        return (None, None, None) + grads


def training_step(model, batch, batch_idx):
    """models.py:256-277."""
    x, y, _ = batch
    eng = get_engine(model)
    x = x.contiguous().float()
    y = y.to(x.device)
    model.train() if not model.training else None
    logits, embeds = eng.forward(x)
    dlogits = torch.empty_like(logits)
    la = ops.softmax_ce(logits, y.contiguous(), dlogits, 1.0 / x.shape[0])
    loss, acc = la[0], la[1]
    model.log_dict({"train_accuracy": acc, "train_loss": loss}, on_step=False, on_epoch=True, prog_bar=True)
    max_epochs = getattr(getattr(model, "trainer", None), "max_epochs", None)
    if max_epochs is not None and model.current_epoch > int(max_epochs / 2):
        y_hat = torch.max(logits, 1).indices
        mask = (y == 0) & (y_hat == 0)
        model.memory_bank = torch.cat([model.memory_bank, embeds[mask].detach().to('cpu')])
    if torch.is_grad_enabled():
        params = [p for p in model.parameters()]
        return _HipLoss.apply(model, loss, dlogits, *params)
    return loss


# ---------------------------------------------------------------------------------------------
# optimiser / scheduler / data-parallel step
# ---------------------------------------------------------------------------------------------
class FusedSGD:
    """torch.optim.SGD(params, lr, momentum, weight_decay) semantics over the flat arena (models.py:337)."""

    def __init__(self, model, lr, momentum=0.9, weight_decay=0.0005):
        self.model, self.momentum, self.weight_decay = model, momentum, weight_decay
        self.param_groups = [{"lr": lr, "initial_lr": lr, "momentum": momentum, "weight_decay": weight_decay}]
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none=False):
        pass        # every gradient range is overwritten (never accumulated) by backward

    def step(self, closure=None):
        if closure is not None:
            closure()
        a = get_engine(self.model).arena
        a.sync_grads()
        lr = self.param_groups[0]["lr"]
        for s, e in a.trainable_ranges():
            ops.sgd_step(a.p[s:e], a.g[s:e], a.m[s:e], lr, self.momentum, self.weight_decay, self.grad_scale)
        self.model._plan = None       # the raw-pointer update does not bump tensor versions: drop folded eval weights

    def state_dict(self):
        """Momentum as ONE flat tensor in backward order, WITHOUT the arena's alignment pads (the layout rounds 1-5 wrote)."""
        a = get_engine(self.model).arena
        return {"param_groups": self.param_groups,
                "momentum": torch.cat([a.m[off:off + n] for off, n in (a.offset[id(p)] for p in a.params)]).cpu()}

    def load_state_dict(self, sd):
        self.param_groups = sd["param_groups"]
        a = get_engine(self.model).arena
        mom = sd["momentum"].to(a.m.device)
        if mom.numel() == a.total:                      # a padded arena image
            a.m.copy_(mom)
            return
        assert mom.numel() == sum(n for _, n in (a.offset[id(p)] for p in a.params)), "momentum does not match this model's parameters"
        o = 0
        for p in a.params:
            off, n = a.offset[id(p)]
            a.m[off:off + n].copy_(mom[o:o + n])
            o += n


class CosineWarmRestarts:
    """CosineAnnealingWarmRestarts(optimizer, T_0) stepped once per epoch (models.py:338)."""

    def __init__(self, optimizer, T_0, eta_min=0.0):
        self.opt, self.T_0, self.eta_min, self.epoch = optimizer, T_0, eta_min, 0
        self.base = optimizer.param_groups[0]["lr"]

    def step(self):
        self.epoch += 1
        t = self.epoch % self.T_0
        self.opt.param_groups[0]["lr"] = self.eta_min + (self.base - self.eta_min) * (1 + math.cos(math.pi * t / self.T_0)) / 2

    def get_last_lr(self):
        return [self.opt.param_groups[0]["lr"]]


def precision_mode(precision):
    """Trainer(precision=...) -> operand mode of the conv / linear kernels.
    32: exact fp32 MFMA.  16 / "16-mixed" (what the reference passes, tools.py:263 = fp16 autocast): fp16 operands,
    fp32 accumulate, dynamic loss scaling.  "bf16" / "bf16-mixed": bf16 operands (explicit opt-in; narrower mantissa than
    the reference's).  "bf16x6" / "bf16x3": split-bf16 emulation of the fp32 product (x6: fp32-faithful, forward + dgrad,
    wgrad stays exact; x3: 4.6e-6, all three contractions)."""
    s = str(precision)
    if s in ("bf16x6", "32x6"):
        return 6
    if s in ("bf16x3", "32x3"):
        return 3
    if s in ("bf16", "bf16-mixed"):
        return True
    if s in ("16", "16-mixed", "fp16"):
        return 2
    if s in ("32", "32-true", "fp32"):
        return False
    raise ValueError(f"unknown precision {precision!r}")


class GradBucketer:
    """Bucketed, overlapped gradient all-reduce over contiguous ranges of a flat gradient arena.

    ``notify(end)`` is called by backward whenever the arena prefix [0, end) has become final.  Trainable ranges
    inside the not-yet-reduced part of that prefix are all-reduced (sum, async) once at least ``min_bucket``
    floats are pending, or when ``final`` is set.  Pure torch.distributed: works over RCCL on GPUs and over gloo
    on CPU tensors (tests/test_distributed_gloo.py).  ``recorder`` (graph capture): an object with ``bucket(lo, hi)``
    and ``wait()`` that is told about the buckets instead of torch.distributed -- the bucket arithmetic stays in one place."""

    def __init__(self, grad_arena, ranges, process_group=None, min_bucket=1 << 20):
        self.g, self.ranges, self.pg, self.min_bucket = grad_arena, list(ranges), process_group, min_bucket
        self.works, self.done, self.launched = [], 0, []
        self.recorder = None

    def reset(self, ranges=None):
        if ranges is not None:
            self.ranges = list(ranges)
        self.works, self.done, self.launched = [], 0, []

    def notify(self, end, final=False):
        if end <= self.done or (end - self.done < self.min_bucket and not final):
            return
        for s, e in self.ranges:
            lo, hi = max(s, self.done), min(e, end)
            if hi > lo:
                if self.recorder is not None:
                    self.recorder.bucket(lo, hi)
                else:
                    self.works.append(dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                self.launched.append((lo, hi))
        self.done = end

    def wait(self):
        if self.recorder is not None:
            self.recorder.wait()
            return
        for w in self.works:
            w.wait()
        self.works = []


def broadcast_replica_state(model, arena, process_group=None, src=0):
    """Make every rank start from rank `src`'s replica: parameters, momentum and the BatchNorm buffers (running
    statistics, num_batches_tracked) -- what DistributedDataParallel does at construction (parameters + buffers) plus
    the optimizer state this engine owns.  The reference is single-device (tools.py:266); replicas that merely seeded
    alike would drift apart silently the first time one of them is built differently."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        dist.broadcast(arena.p, src=src, group=process_group)
        dist.broadcast(arena.m, src=src, group=process_group)
        for b in model.buffers():
            dist.broadcast(b, src=src, group=process_group)


def bit_checksum(t):
    """Order-independent, exact fingerprint of a float tensor's BITS: int64 [sum of the int32 words, sum of their squares' low
    halves, count of non-finite-looking exponent fields].  Equal tensors give equal checksums on every rank; a single flipped
    bit changes the first entry."""
    w = t.detach().contiguous().view(torch.int32).to(torch.int64)
    return torch.stack([w.sum(), (w & 0xFFFF).mul_(w >> 16 & 0xFFFF).sum(), ((w >> 23 & 0xFF) == 0xFF).sum()])


def ranks_agree(checksum, process_group=None):
    """True when every rank holds the same checksum vector (one MAX and one MIN all-reduce of a few int64: gloo or RCCL)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return True
    hi, lo = checksum.clone(), checksum.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=process_group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=process_group)
    return bool(torch.equal(hi.cpu(), lo.cpu()))


def all_ranks_true(flag, device, process_group=None):
    """Collective AND of a per-rank boolean."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], device=device, dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=process_group)
    return bool(int(t.item()) == 1)


class LossScaler:
    """torch.cuda.amp.GradScaler's state machine (what PL runs for the reference's precision=16, tools.py:263) kept in
    three device floats [scale, growth_tracker, found_inf] so that no step needs a host round trip."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.state = torch.tensor([init_scale, 0.0, 0.0], device=device, dtype=torch.float32)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval

    def get_scale(self):
        return float(self.state[0])


class StepWatchdog:
    """Stall detector of a replayed multi-rank step.  `_replay` records a HIP event behind every plan op (graph segment, all-reduce,
    wait); a daemon thread looks at the newest issued step once a second, and when its last event is still pending `bound_s` seconds
    after the step was issued it reports the FIRST op whose event has not completed -- `[ssad watchdog] rank R: step S stalled
    > B s at plan op i of n (kind)` on stderr -- and ends the process with a non-zero status (os._exit: a rank that hangs inside a
    collective cannot be unwound; torchrun then takes the other ranks down, so every rank exits non-zero).  It never re-execs
    anything; a launcher may start a fresh job with SSAD_GRAPH=0.  Why: hipGraph segments interleaved with RCCL all-reduces first
    meet on the driver's 8-GPU run, and a silent hang there would say nothing about where it stopped.
    Bound: SSAD_STEP_TIMEOUT_S (default 120 s; 0 switches the watchdog off).  `on_stall` replaces the exit (tests)."""

    EXIT_CODE = 87

    def __init__(self, rank=0, bound_s=None, on_stall=None, poll_s=1.0):
        import threading
        self.rank = rank
        self.bound_s = float(os.environ.get("SSAD_STEP_TIMEOUT_S", "120")) if bound_s is None else float(bound_s)
        self.on_stall, self.poll_s = on_stall, poll_s
        self._lock = threading.Lock()
        self._pending = []            # issued steps not yet seen complete, oldest first: (step index, issue time, [(kind, event)])
        self._steps = 0
        self._paused = False
        self._thread = None

    def enabled(self):
        return self.bound_s > 0

    def arm(self, ops_events):
        """Called by the issuing thread once a step's launches are out: ops_events = [(kind, event with .query())]."""
        import threading
        import time
        if not self.enabled():
            return
        with self._lock:
            self._steps += 1
            self._pending.append((self._steps, time.monotonic(), list(ops_events)))
            if len(self._pending) > 256:          # (the issuing thread never runs this far ahead of the device; a bound all the same)
                del self._pending[:-256]
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, name="ssad-step-watchdog", daemon=True)
            self._thread.start()

    def pause(self, flag):
        """No event queries while the issuing thread records a graph (stream capture)."""
        with self._lock:
            self._paused = bool(flag)

    def check(self, now=None):
        """One look at the OLDEST unfinished step (the issuing thread runs ahead of the device: a stall belongs to the first step whose
        events stop completing) -> None, or (step, op index, n ops, kind) of its first pending op once it is overdue."""
        import time
        with self._lock:
            if self._paused:
                return None
            while self._pending and (not self._pending[0][2] or self._pending[0][2][-1][1].query()):
                self._pending.pop(0)              # finished steps leave the queue
            cur = self._pending[0] if self._pending else None
        if cur is None:
            return None
        step, t0, evs = cur
        now = time.monotonic() if now is None else now
        if now - t0 <= self.bound_s:
            return None
        for i, (kind, ev) in enumerate(evs):
            if not ev.query():
                return step, i, len(evs), kind
        return None

    def _run(self):
        import sys
        import time
        while True:
            time.sleep(self.poll_s)
            hit = self.check()
            if hit is None:
                continue
            step, i, n, kind = hit
            msg = (f"[ssad watchdog] rank {self.rank}: step {step} stalled > {self.bound_s:g} s at plan op {i} of {n} ({kind}); "
                   f"exiting with status {self.EXIT_CODE} (relaunch with SSAD_GRAPH=0 for eager launches)")
            print(msg, file=sys.stderr, flush=True)
            if self.on_stall is not None:
                self.on_stall(hit)
                with self._lock:
                    self._pending = []
                continue
            os._exit(self.EXIT_CODE)


class DataParallelStep:
    """forward + loss + backward + (bucketed RCCL all-reduce overlapped with backward) + SGD for one rank.

    Buckets are the arena prefixes that become final after the head, after each ResNet stage's first block, and
    after the stem; each is all-reduced (sum) as soon as it is final, on RCCL's stream, while the compute stream
    continues with the next stage.  The 1/world factor is applied inside the SGD kernel.  BatchNorm statistics
    stay per rank during training (the reference has no SyncBN); rank 0's replica (parameters, momentum, BatchNorm
    buffers) is broadcast once at construction, as DistributedDataParallel does.

    hipGraph replay (``graph=True``, the default; SSAD_GRAPH=0 turns it off): the ~300 launches of a step are captured
    once per input shape into graph SEGMENTS that end wherever a gradient bucket becomes final; a step is then
    segment, all-reduce(bucket), segment, ... , wait, SGD segment.  The collectives themselves are never captured (they
    are issued eagerly between the segment launches, on RCCL's own stream), so the same plan serves 1 and N ranks.
    Learning rate and loss scale live in device memory: a captured step never bakes them in."""

    def __init__(self, model, lr, momentum=0.9, weight_decay=0.0005, world_size=None, process_group=None, precision=32,
                 graph=None):
        self.model = model
        self.eng = get_engine(model)
        self.eng.bf16 = precision_mode(precision)
        self.opt = FusedSGD(model, lr, momentum, weight_decay)
        self.world = world_size if world_size is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.opt.grad_scale = 1.0 / self.world
        self.pg = process_group
        a = self.eng.arena
        self.bucketer = GradBucketer(a.g, a.trainable_ranges(), process_group)
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            broadcast_replica_state(model, a, process_group)
        self.scaler = LossScaler(a.p.device) if self.eng.bf16 == 2 else None      # fp16 operands: dynamic loss scaling
        self.hyper = torch.zeros(4, device=a.p.device, dtype=torch.float32)
        self._hyper_host = None
        import os
        self.use_graph = (os.environ.get("SSAD_GRAPH", "1") != "0") if graph is None else bool(graph)
        self._plans, self._seen = {}, {}
        self._cap_stream = None
        self.launch_mode = "hipGraph segments" if self.use_graph else "eager"
        self.self_check_report = None
        self.comm_events = None       # bench.py: a list -> (event, event) pairs around every wait for the gradient all-reduce
        # multi-rank replays are watched (StepWatchdog); SSAD_WATCHDOG=1 forces it on one rank (tests), =0 switches it off
        wd = os.environ.get("SSAD_WATCHDOG", "")
        rank = dist.get_rank(process_group) if (dist.is_available() and dist.is_initialized()) else 0
        self.watchdog = StepWatchdog(rank) if ((self.world > 1 and wd != "0") or wd == "1") else None

    # ---- pieces ----
    def _sync_hyper(self):
        h = (float(self.opt.param_groups[0]["lr"]), float(self.opt.momentum), float(self.opt.weight_decay),
             float(self.opt.grad_scale))
        if h != self._hyper_host:
            self.hyper.copy_(torch.tensor(h, dtype=torch.float32), non_blocking=False)
            self._hyper_host = h

    def _notify(self, end):
        a = self.eng.arena
        final = end == a.total or (not self.eng.trunk_grad and end == a.head_end)
        self.bucketer.notify(end, final)

    def _body(self, x, y):
        """The launches of one step, in order.  Returns (loss/acc, logits, embeddings)."""
        eng = self.eng
        a = eng.arena
        ops.ASIDE = eng.aside if eng.sw_aside else None
        ops.PENDING_REDUCE = [] if eng.sw_batch_reduce else None
        try:
            logits, emb = eng.forward(x)
            dlogits = torch.empty_like(logits)
            la = ops.softmax_ce(logits, y, dlogits, 1.0 / x.shape[0])
            sc = self.scaler.state if self.scaler is not None else None
            if sc is not None:
                ops.scale_by_loss_scale(dlogits, sc)
            eng.backward(dlogits)
        finally:
            eng.aside.join()                  # (also on an exception: a capture must not end with an un-joined branch)
            ops.ASIDE = None
            ops.PENDING_REDUCE = None
        if self.bucketer.recorder is None and self.bucketer.works:      # eager launches: the same (optionally timed) wait as a replay
            self._wait_works(self.bucketer.works)
            self.bucketer.works = []
        else:
            self.bucketer.wait()
        rng = a.trainable_ranges()
        if sc is not None:
            for s, e in rng:
                ops.check_finite(a.g[s:e], sc)
        for s, e in rng:
            ops.sgd_step_dev(a.p[s:e], a.g[s:e], a.m[s:e], self.hyper, sc)
        if sc is not None:
            ops.loss_scaler_update(sc, self.scaler.growth_factor, self.scaler.backoff_factor, self.scaler.growth_interval)
        return la, logits, emb

    # ---- eager ----
    def _step_eager(self, x, y):
        self.eng.bucket_hooks = self._notify if self.world > 1 else None
        self.bucketer.reset(self.eng.arena.trainable_ranges())
        return self._body(x, y)

    # ---- captured ----
    def _capture(self, x, y, key):
        """Record the step for this input shape as graph segments cut at the gradient-bucket boundaries."""
        plan = {"x": torch.empty_like(x).copy_(x), "y": torch.empty_like(y).copy_(y), "ops": []}
        pool = torch.cuda.graph_pool_handle()
        if self._cap_stream is None:               # ONE side stream for every capture of this object (streams are multiplexed
            self._cap_stream = torch.cuda.Stream()  # onto a few hardware queues: a new one per input shape is not free)
        stream = self._cap_stream
        multi = self.world > 1
        seg = {"g": None, "mark": 0}

        def begin():
            seg["g"] = torch.cuda.CUDAGraph()
            seg["mark"] = _hip.LAUNCHES
            # thread-local capture mode: RCCL's watchdog thread may query events while this thread captures (a "global" capture
            # would be invalidated by any such call from another thread)
            seg["g"].capture_begin(pool=pool, capture_error_mode="thread_local")

        def end():
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")            # an empty segment (two buckets back to back) is dropped below
                seg["g"].capture_end()
            if _hip.LAUNCHES != seg["mark"]:
                plan["ops"].append(("graph", seg["g"]))

        class _Recorder:          # a final bucket / the wait before SGD ends the current segment
            def bucket(_, lo, hi):
                end()
                plan["ops"].append(("allreduce", lo, hi))
                begin()

            def wait(_):
                if multi:
                    end()
                    plan["ops"].append(("wait",))
                    begin()

        self.eng.bucket_hooks = self._notify if multi else None
        self.bucketer.reset(self.eng.arena.trainable_ranges())
        self.bucketer.recorder = _Recorder()
        stream.wait_stream(torch.cuda.current_stream())
        if self.watchdog is not None:
            self.watchdog.pause(True)
        try:
            with torch.cuda.stream(stream):
                begin()
                try:
                    plan["out"] = self._body(plan["x"], plan["y"])
                except BaseException:
                    try:                                   # leave capture mode, but let the ORIGINAL error propagate
                        seg["g"].capture_end()
                    except Exception:
                        pass
                    raise
                end()
        finally:
            self.bucketer.recorder = None
            if self.watchdog is not None:
                self.watchdog.pause(False)
        torch.cuda.current_stream().wait_stream(stream)
        self._plans[key] = plan
        return plan

    def _replay(self, plan, x, y):
        if x.data_ptr() != plan["x"].data_ptr():
            plan["x"].copy_(x, non_blocking=True)
        if y.data_ptr() != plan["y"].data_ptr():
            plan["y"].copy_(y, non_blocking=True)
        works = []
        g = self.eng.arena.g
        wd = self.watchdog if (self.watchdog is not None and self.watchdog.enabled()) else None
        evs = None
        if wd is not None:
            # one event per plan op, in a ring of four sets: a set is re-recorded only after the step that last used it has finished
            # (the issuing thread stays at most four steps ahead of the device), so a pending event always belongs to the step the
            # watchdog attributes it to
            if "events" not in plan:
                plan["events"], plan["ev_step"] = [[torch.cuda.Event() for _ in plan["ops"]] for _ in range(4)], 0
            evs = plan["events"][plan["ev_step"] % 4]
            if plan["ev_step"] >= 4:
                evs[-1].synchronize()
            plan["ev_step"] += 1
        for i, op in enumerate(plan["ops"]):
            if op[0] == "graph":
                op[1].replay()
            elif op[0] == "allreduce":
                works.append(dist.all_reduce(g[op[1]:op[2]], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            else:
                self._wait_works(works)
                works = []
            if wd is not None:
                # (an all-reduce runs on RCCL's stream: its event marks the compute stream's position when it was issued; a stalled
                # collective shows up at the "wait" op that follows it)
                evs[i].record()
        if wd is not None:
            kinds = [f"graph segment {sum(1 for o in plan['ops'][:i + 1] if o[0] == 'graph')}" if op[0] == "graph" else
                     (f"all-reduce of g[{op[1]}:{op[2]}]" if op[0] == "allreduce" else "wait for the all-reduces")
                     for i, op in enumerate(plan["ops"])]
            wd.arm(list(zip(kinds, evs)))
        return plan["out"]

    def _wait_works(self, works):
        """The compute stream waits for the outstanding all-reduces.  With ``comm_events`` set (bench.py's probe) the wait is bracketed by
        HIP events on the compute stream: nothing else is queued between them, so their distance is the time the step stands still
        for the exchange -- the part of the all-reduce that backward did NOT hide."""
        ev = getattr(self, "comm_events", None)
        if ev is None:
            for w in works:
                w.wait()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for w in works:
            w.wait()
        e1.record()
        ev.append((e0, e1))

    def allreduce_bytes(self):
        """Gradient bytes one rank hands to the all-reduce per step (fp32; the trainable arena ranges)."""
        return 4 * sum(e - s for s, e in self.eng.arena.trainable_ranges())

    def allreduce_messages(self):
        plan = next(iter(self._plans.values()), None)
        if plan is not None:
            return sum(1 for o in plan["ops"] if o[0] == "allreduce")
        return len(self.bucketer.launched)

    # ---- start-up self-check of a multi-rank job ----
    def _snapshot(self):
        a = self.eng.arena
        return {"p": a.p.clone(), "m": a.m.clone(), "buf": [b.clone() for b in self.model.buffers()],
                "sc": self.scaler.state.clone() if self.scaler is not None else None}

    def _restore(self, snap):
        a = self.eng.arena
        with torch.no_grad():
            a.p.copy_(snap["p"]); a.m.copy_(snap["m"])
            for b, v in zip(self.model.buffers(), snap["buf"]):
                b.copy_(v)
            if snap["sc"] is not None:
                self.scaler.state.copy_(snap["sc"])
        self.model._plan = None

    def self_check(self, x, y):
        """Run ONE step on (x, y) twice from the same state -- launched eagerly, then recorded as hipGraph segments and replayed
        (collectives eagerly between the segments) -- and compare: (i) on this rank the two must leave bit-identical parameters,
        momentum and BatchNorm buffers; (ii) after either, every rank must hold the same parameters and momentum (exact bit checksum,
        MAX / MIN all-reduce).  The state is put back afterwards, so the check does not move the training trajectory.  If the replay
        disagrees with the eager step on ANY rank, or the replicas disagree after the replayed step, every rank falls back to
        eager launches (collective decision) and says so in `launch_mode`; replicas that disagree after the EAGER step are
        re-synchronised from rank 0 and reported (`replicas_agree_eager: false`): that is a defect no fallback hides.
        Why: hipGraph segments + RCCL across several GPUs first meet on the driver's 8-GPU run (the box this is built on has
        one GPU); a silently diverged model must not produce a scaling figure.  -> dict for the bench line / logs."""
        dev = self.eng.arena.p.device
        rep = {"world": self.world, "graph_requested": bool(self.use_graph)}
        self._sync_hyper()
        snap = self._snapshot()
        self._step_eager(x, y)
        torch.cuda.synchronize(dev)
        a = self.eng.arena
        state = lambda: torch.cat([a.p, a.m] + [b.detach().flatten().float() for b in self.model.buffers()])
        # what must be identical ACROSS ranks: parameters and momentum (BatchNorm running statistics are per rank by design: each
        # rank normalises its own shard, the reference has no SyncBN)
        shared = lambda: bit_checksum(torch.cat([a.p, a.m]))
        eager = state().clone()
        rep["replicas_agree_eager"] = ranks_agree(shared(), self.pg)
        self._restore(snap)
        ok_local = True
        if self.use_graph:
            key = self._plan_key(x, y)
            # Phase 1, no collective inside: every rank records its plan (the recorder notes the buckets, it does not reduce them).
            plan = None
            try:
                plan = self._capture(x, y, key)
            except Exception as e:           # noqa: BLE001  (a capture the runtime refuses is a reason to run eagerly, not to stop)
                rep["capture_error"] = f"{type(e).__name__}: {e}"
                self._plans.pop(key, None)
            # Phase 2: ONE collective decision.  A rank that failed to capture must not meet ranks that go on to issue the replay's
            # per-bucket all-reduces with a different collective (mismatched RCCL collectives hang or corrupt): either every rank
            # replays, or none does and all of them fall back together.
            captured_everywhere = all_ranks_true(plan is not None, dev, self.pg)
            rep["captured_on_every_rank"] = captured_everywhere
            ok_local = False
            if captured_everywhere:
                try:
                    self._replay(plan, x, y)
                    torch.cuda.synchronize(dev)
                except Exception as e:
                    # part of the collective sequence may be out already: no fallback can realign the ranks from here
                    raise RuntimeError("self-check: the replayed step failed after its collectives had started; "
                                       "aborting the job (run with SSAD_GRAPH=0 for eager launches)") from e
                replay = state()
                ok_local = bool(torch.equal(eager, replay))
                rep["replicas_agree_replay"] = ranks_agree(shared(), self.pg)
                rep["graph_segments"] = sum(1 for o in plan["ops"] if o[0] == "graph")
            else:
                rep["replicas_agree_replay"] = False
                self._plans.pop(key, None)
            self._restore(snap)
            rep["graph_equals_eager"] = all_ranks_true(ok_local, dev, self.pg)
            if not (rep["graph_equals_eager"] and rep.get("replicas_agree_replay", True)):
                self.use_graph = False
                self._plans.clear()
        if not rep["replicas_agree_eager"]:
            broadcast_replica_state(self.model, self.eng.arena, self.pg)
        self.launch_mode = "hipGraph segments" if self.use_graph else (
            "eager" if not rep["graph_requested"] else "eager (self-check: the replayed step did not reproduce the eager one)")
        rep["launch_mode"] = self.launch_mode
        self.self_check_report = rep
        return rep

    def bind_inputs(self, x, y):
        """-> (x, y) tensors a producer may fill IN PLACE for the next step: the recorded plan's own input buffers for this shape
        (step(x, y) then replays without the device-to-device copy of the batch -- 201 MB per step at 256 x 3 x 256 x 256); the given
        tensors themselves while no plan exists for the shape (eager launches, first steps)."""
        plan = self._plans.get(self._plan_key(x, y)) if self.use_graph else None
        if plan is None:
            return x, y
        if plan["x"].data_ptr() != x.data_ptr():
            plan["x"].copy_(x)
            plan["y"].copy_(y)
        return plan["x"], plan["y"]

    def _plan_key(self, x, y):
        return (tuple(x.shape), tuple(y.shape), x.dtype, y.dtype, self.eng.switches(), tuple(self.eng.arena.trainable_ranges()),
                tuple(bool(m.training) for m in self.model.modules() if isinstance(m, BN_TYPES)))

    def step(self, x, y):
        self._sync_hyper()
        key = self._plan_key(x, y)
        if self.use_graph and ops.PROFILE is None:
            plan = self._plans.get(key)
            if plan is None:
                n = self._seen.get(key, 0)
                self._seen[key] = n + 1
                if n >= 1:              # first step of a shape runs eagerly (lazy kernel attributes, communicators)
                    plan = self._capture(x, y, key)
            if plan is not None:
                la, logits, emb = self._replay(plan, x, y)
                self.model._plan = None
                self.last_logits, self.last_embeddings = logits, emb
                return la
        la, logits, emb = self._step_eager(x, y)
        self.model._plan = None       # folded eval-mode weights are stale now (the kernels do not bump tensor versions)
        self.last_logits, self.last_embeddings = logits, emb
        return la
