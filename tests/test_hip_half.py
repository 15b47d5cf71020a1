"""Half-tensor forms of the precision-16 training step (`_h` entry points of include/ssad.h).

pl.Trainer(precision=16) (tools.py:263 of the reference) keeps every trunk activation and its gradient as an fp16 tensor.  Each `_h`
kernel is the SAME kernel as its fp32-tensor namesake with the tensors read / written as halves, so on inputs that are
representable as halves the two must agree to the last bit once the fp32-tensor result is rounded -- that identity is what is
tested here, kernel by kernel; the arithmetic itself is pinned by the fp32-tensor tests (test_hip_training.py) against torch."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _r(t):
    """Round to fp16 and back: a value a half tensor can hold."""
    return t.half().float()


def _bn_params(c, g, dev):
    mean, invstd = (torch.randn(c, generator=g) * 0.2).to(dev), (torch.rand(c, generator=g) + 0.5).to(dev)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(dev), (torch.randn(c, generator=g) * 0.3).to(dev)
    return mean, invstd, gamma, beta


def test_weight_copies(dev):
    from self_supervised import _hip, ops
    import ctypes
    g = torch.Generator().manual_seed(1)
    src = torch.randn(1000 * 8 + 5, generator=g).to(dev)
    dst = torch.empty(src.numel(), device=dev, dtype=torch.float16)
    ops.cvt_f32_f16(src, dst)
    assert torch.equal(dst, src.half())
    # two filters of one arena, flipped as halves == the fp32 flip, rounded
    w1, w2 = torch.randn(96, 3, 3, 64, generator=g), torch.randn(128, 1, 1, 64, generator=g)
    arena = torch.cat([w1.flatten(), w2.flatten()]).to(dev)
    desc = [0, 0, 96, 64, 3, 3, w1.numel(), w1.numel(), 128, 64, 1, 1]
    out = torch.empty(arena.numel(), device=dev, dtype=torch.float16)
    _hip.check(_hip.lib().ssad_flip_transpose_batch_h(_hip.ptr(arena), out.data_ptr(), (ctypes.c_int64 * 12)(*desc), 2, _hip.stream()))
    f1 = ops.flip_transpose_weight(w1.to(dev)).half()
    f2 = ops.flip_transpose_weight(w2.to(dev)).half()
    assert torch.equal(out[:w1.numel()].view(f1.shape), f1) and torch.equal(out[w1.numel():].view(f2.shape), f2)


@pytest.mark.parametrize("shape", [(3, 8, 8, 64, 64, 3, 1, 1), (2, 9, 9, 64, 128, 3, 2, 1), (4, 6, 6, 128, 256, 3, 1, 1),
                                   (2, 8, 8, 64, 128, 1, 2, 0), (64, 32, 32, 128, 128, 3, 1, 1), (5, 7, 5, 256, 512, 3, 2, 1)])
def test_conv_igemm_half_tensors(dev, shape):
    """conv + statistics, input gradient (+ residual), weight gradient: half tensors == the fp16-operand kernels on fp32 tensors
    holding the same (half-representable) values, rounded once; ragged tiles, both strides, a >= 500-workgroup grid."""
    from self_supervised import ops
    n, h, w, cin, cout, k, s, p = shape
    g = torch.Generator().manual_seed(n * 13 + k + cin)
    x = _r(torch.randn(n, h, w, cin, generator=g)).to(dev)
    wt = _r(torch.randn(cout, k, k, cin, generator=g) / (cin * k * k) ** 0.5).to(dev)
    rm1, rv1 = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    rm2, rv2 = rm1.clone(), rv1.clone()
    z32, m32, i32 = ops.conv_fwd_stats(x, wt, 1e-5, 0.1, rm1, rv1, s, p, bf16=2)
    zh, mh, ih = ops.conv_fwd_stats(x.half(), wt.half(), 1e-5, 0.1, rm2, rv2, s, p, bf16=2)
    assert zh.dtype == torch.float16 and torch.equal(zh, z32.half())
    # statistics are those of the STORED halves
    zc = zh.double().reshape(-1, cout)
    assert (mh.double() - zc.mean(0)).abs().max().item() < 1e-5 * max(1.0, zc.abs().max().item())
    assert ((ih.double() - (zc.var(0, unbiased=False) + 1e-5).rsqrt()).abs() / ih.double()).max().item() < 1e-5
    assert (mh - m32).abs().max().item() < 2e-3 and ((ih - i32).abs() / i32).max().item() < 2e-3
    # input gradient
    dy = _r(torch.randn(z32.shape, generator=g)).to(dev)
    res = _r(torch.randn(x.shape, generator=g)).to(dev)
    wf = ops.flip_transpose_weight(wt)
    dx32 = ops.conv_dgrad(dy, wf, x.shape, s, p, res, bf16=2)
    dxh = ops.conv_dgrad(dy.half(), wf.half(), x.shape, s, p, res.half(), bf16=2)
    assert dxh.dtype == torch.float16 and torch.equal(dxh, dx32.half())
    dxh = ops.conv_dgrad(dy.half(), wf.half(), x.shape, s, p, None, bf16=2)
    assert torch.equal(dxh, ops.conv_dgrad(dy, wf, x.shape, s, p, None, bf16=2).half())
    # weight gradient (fp32 either way)
    dw32 = torch.empty(cout * k * k * cin, device=dev)
    dwh = torch.empty_like(dw32)
    ops.conv_wgrad(dy, x, dw32, k, k, s, p, bf16=2)
    ops.conv_wgrad(dy.half(), x.half(), dwh, k, k, s, p, bf16=2)
    # 1 x 1 layers: the same kernel reading halves (equal bits); 3 x 3 layers: csrc/wgrad16.hip -- the same products, another order
    if k == 1:
        assert torch.equal(dwh, dw32)
    else:
        assert (dwh - dw32).abs().max().item() <= 2e-5 * max(1.0, dw32.abs().max().item())


@pytest.mark.parametrize("shape", [(3, 16, 16), (2, 13, 21), (5, 64, 64)])
def test_conv_c64_half_tensors(dev, shape):
    """Halo-tile 64 -> 64 conv with half tensors: plain, residual, producer BatchNorm + ReLU on load with the emitted activation and the
    output statistics; as input gradient; ragged tiles."""
    from self_supervised import ops
    n, h, w = shape
    g = torch.Generator().manual_seed(n * 10 + h)
    x = _r(torch.randn(n, h, w, 64, generator=g)).to(dev)
    wt = (torch.randn(64, 3, 3, 64, generator=g) / 24.0).to(dev)          # fp32 master weights, rounded while staged in both forms
    res = _r(torch.randn(n, h, w, 64, generator=g)).to(dev)
    assert torch.equal(ops.conv3x3_c64(x.half(), wt, bf16=2), ops.conv3x3_c64(x, wt, bf16=2).half())
    assert torch.equal(ops.conv3x3_c64(x.half(), wt, residual=res.half(), bf16=2), ops.conv3x3_c64(x, wt, residual=res, bf16=2).half())
    tr = _bn_params(64, g, dev)
    rm1, rv1 = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    rm2, rv2 = rm1.clone(), rv1.clone()
    z32, e32, m32, i32 = ops.conv3x3_c64(x, wt, transform=tr, emit=True, stats=(1e-5, 0.1, rm1, rv1), bf16=2)
    zh, eh, mh, ih = ops.conv3x3_c64(x.half(), wt, transform=tr, emit=True, stats=(1e-5, 0.1, rm2, rv2), bf16=2)
    assert torch.equal(eh, e32.half()) and torch.equal(zh, z32.half())
    zc = zh.double().reshape(-1, 64)
    assert (mh.double() - zc.mean(0)).abs().max().item() < 1e-5 * max(1.0, zc.abs().max().item())
    assert ((ih.double() - (zc.var(0, unbiased=False) + 1e-5).rsqrt()).abs() / ih.double()).max().item() < 1e-5
    assert (rm2 - rm1).abs().max().item() < 1e-3


@pytest.mark.parametrize("shape", [(3, 16, 16, 64, 64), (2, 13, 21, 64, 128), (5, 8, 8, 128, 256), (3, 5, 7, 256, 64), (7, 8, 8, 512, 512),
                                   (40, 32, 32, 128, 128), (3, 64, 64, 64, 64), (600, 8, 8, 64, 64)])
def test_conv3x3_h_halo_kernel(dev, shape):
    """csrc/conv16.hip against the implicit GEMM over the same half tensors (same products, fp32 accumulation in a different order):
    plain + statistics, residual (the input-gradient use), producer BatchNorm + ReLU on load with the emitted activation; 8 x 16 tiles
    and the two-maps-per-tile form of maps up to 8 wide, ragged tiles, odd sub-tile counts, one and several channel slabs, persistent
    workgroups that walk many tiles (the 600-map case: > 512 tiles)."""
    from self_supervised import ops
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(n * 3 + h + cin)
    x = torch.randn(n, h, w, cin, generator=g).half().to(dev)
    wt = (torch.randn(cout, 3, 3, cin, generator=g) / (9 * cin) ** 0.5).half().to(dev)
    res = torch.randn(n, h, w, cout, generator=g).half().to(dev)
    rm1, rv1 = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    rm2, rv2 = rm1.clone(), rv1.clone()
    zi, mi, ii = ops.conv_fwd_stats(x, wt, 1e-5, 0.1, rm1, rv1, 1, 1, bf16=2)
    z, m, i = ops.conv3x3_h(x, wt, stats=(1e-5, 0.1, rm2, rv2))
    tol = lambda ref: 2e-3 * max(1.0, ref.float().abs().max().item())               # one half ulp of the largest value, twice
    assert z.dtype == torch.float16 and (z.float() - zi.float()).abs().max().item() <= tol(zi)
    zc = z.double().reshape(-1, cout)
    assert (m.double() - zc.mean(0)).abs().max().item() < 1e-5 * max(1.0, zc.abs().max().item())
    assert ((i.double() - (zc.var(0, unbiased=False) + 1e-5).rsqrt()).abs() / i.double()).max().item() < 1e-5
    assert (rm2 - rm1).abs().max().item() < 2e-3 and (rv2 - rv1).abs().max().item() < 2e-3
    # fp32 math on the same operands
    want = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wt.float().cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    assert (z.float().cpu() - want).abs().max().item() <= tol(want)
    # residual
    zr = ops.conv3x3_h(x, wt, residual=res)
    assert (zr.float().cpu() - (want + res.float().cpu())).abs().max().item() <= tol(want + res.float().cpu())
    # producer BatchNorm + ReLU on load == bn_apply_fwd over half tensors, then the plain kernel
    tr = _bn_params(cin, g, dev)
    act = ops.bn_apply_fwd(x, tr[0], tr[1], tr[2], tr[3], None, True)
    z2, em, m2, i2 = ops.conv3x3_h(x, wt, transform=tr, emit=True, stats=(1e-5, 0.1, rm2, rv2))
    assert torch.equal(em, act)                                                     # the same expression, rounded once
    z3 = ops.conv3x3_h(act, wt)
    assert torch.equal(z2, z3)


@pytest.mark.parametrize("shape", [(26, 64, 64, 64, 64), (52, 32, 32, 128, 128), (104, 16, 16, 256, 256), (202, 8, 8, 512, 512),
                                   (52, 32, 32, 64, 128), (26, 64, 64, 128, 64), (801, 8, 8, 64, 128), (401, 16, 16, 64, 64)])
def test_conv3x3_hw_register_fed_kernel(dev, shape):
    """csrc/conv16w.hip (filters packed in fragment order and fed from registers, halo staged by two extra waves) against fp32 math on
    the same half operands and against csrc/conv16.hip: 16 x 16 tiles and four 8 x 8 maps per tile (ragged last tile), one / two / four
    channel slabs, 64- and 32-channel chunks, statistics, residual, producer BatchNorm + ReLU on load with the emitted activation; the
    flipped pack == the pack of the flipped filter."""
    from self_supervised import ops
    n, h, w, cin, cout = shape
    assert ops.conv3x3_hw_ok(n, h, w, cin, cout)
    g = torch.Generator().manual_seed(n * 3 + h + cin)
    x = torch.randn(n, h, w, cin, generator=g).half().to(dev)
    w32 = (torch.randn(cout, 3, 3, cin, generator=g) / (9 * cin) ** 0.5).to(dev)
    wt = w32.half()
    res = torch.randn(n, h, w, cout, generator=g).half().to(dev)
    wp, _ = ops.conv3x3_hw_pack(w32.reshape(-1), [(0, cout, cin, False)])
    rm1, rv1 = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    rm2, rv2 = rm1.clone(), rv1.clone()
    zo, mo, io = ops.conv3x3_h(x, wt, stats=(1e-5, 0.1, rm1, rv1))
    z, m, i = ops.conv3x3_hw(x, wp, cout, stats=(1e-5, 0.1, rm2, rv2))
    tol = lambda ref: 2e-3 * max(1.0, ref.float().abs().max().item())               # one half ulp of the largest value, twice
    want = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wt.float().cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    assert z.dtype == torch.float16 and (z.float().cpu() - want).abs().max().item() <= tol(want)
    assert (z.float() - zo.float()).abs().max().item() <= tol(want)
    zc = z.double().reshape(-1, cout)
    assert (m.double() - zc.mean(0)).abs().max().item() < 1e-5 * max(1.0, zc.abs().max().item())
    assert ((i.double() - (zc.var(0, unbiased=False) + 1e-5).rsqrt()).abs() / i.double()).max().item() < 1e-5
    assert (rm2 - rm1).abs().max().item() < 2e-3 and (rv2 - rv1).abs().max().item() < 2e-3
    zr = ops.conv3x3_hw(x, wp, cout, residual=res)
    assert (zr.float().cpu() - (want + res.float().cpu())).abs().max().item() <= tol(want + res.float().cpu())
    tr = _bn_params(cin, g, dev)
    act = ops.bn_apply_fwd(x, tr[0], tr[1], tr[2], tr[3], None, True)
    z2, em, m2, i2 = ops.conv3x3_hw(x, wp, cout, transform=tr, emit=True, stats=(1e-5, 0.1, rm2, rv2))
    assert torch.equal(em, act)                                                     # the same expression, rounded once
    assert torch.equal(z2, ops.conv3x3_hw(act, wp, cout))
    # input gradient: the flipped pack of the forward filter [cout][3][3][cin] runs a conv cout -> cin
    if ops.conv3x3_hw_ok(n, h, w, cout, cin):
        wpf, _ = ops.conv3x3_hw_pack(w32.reshape(-1), [(0, cin, cout, True)])
        wflip = w32.flip(1, 2).permute(3, 1, 2, 0).contiguous()                     # [cin][3][3][cout]
        wpf2, _ = ops.conv3x3_hw_pack(wflip.reshape(-1), [(0, cin, cout, False)])
        assert torch.equal(wpf, wpf2)
        dx = ops.conv3x3_hw(res, wpf, cin)
        wantdx = F.conv_transpose2d(res.float().cpu().permute(0, 3, 1, 2), wt.float().cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
        assert (dx.float().cpu() - wantdx).abs().max().item() <= tol(wantdx)


@pytest.mark.parametrize("shape", [(64, 64, 64, 64, 64), (128, 32, 32, 128, 128), (104, 16, 16, 256, 256), (202, 8, 8, 512, 512),
                                   (128, 32, 32, 64, 128), (64, 64, 64, 128, 64), (1025, 16, 16, 64, 64)])
def test_conv3x3_fw_exact_fp32_form(dev, shape):
    """The float instantiation of csrc/conv16w.hip (v_mfma_f32_32x32x2_f32, statistics in double per value) against F.conv2d in fp32 and
    against the implicit GEMM of the exact-fp32 step: plain + statistics, residual, residual gated by a nibble mask (the identity-branch
    gradient of a residual block), producer BatchNorm + ReLU on load with the emitted activation, the flipped pack as input gradient."""
    from self_supervised import ops
    n, h, w, cin, cout = shape
    assert ops.conv3x3_hw_ok(n, h, w, cin, cout, f32=True)
    g = torch.Generator().manual_seed(n * 7 + h + cin)
    x = torch.randn(n, h, w, cin, generator=g).to(dev)
    w32 = (torch.randn(cout, 3, 3, cin, generator=g) / (9 * cin) ** 0.5).to(dev)
    res = torch.randn(n, h, w, cout, generator=g).to(dev)
    wp, _ = ops.conv3x3_hw_pack(w32.reshape(-1), [(0, cout, cin, False)], f32=True)
    rm1, rv1 = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    rm2, rv2 = rm1.clone(), rv1.clone()
    zi, mi, ii = ops.conv_fwd_stats(x, w32, 1e-5, 0.1, rm1, rv1, 1, 1)
    z, m, i = ops.conv3x3_hw(x, wp, cout, stats=(1e-5, 0.1, rm2, rv2))
    want = F.conv2d(x.cpu().permute(0, 3, 1, 2), w32.cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    tol = 2e-5 * max(1.0, want.abs().max().item())
    assert z.dtype == torch.float32 and (z.cpu() - want).abs().max().item() <= tol
    assert (z - zi).abs().max().item() <= tol
    assert (m - mi).abs().max().item() < 1e-6 and ((i - ii).abs() / ii).max().item() < 1e-6
    assert (rm2 - rm1).abs().max().item() < 1e-6 and (rv2 - rv1).abs().max().item() < 1e-6
    zr = ops.conv3x3_hw(x, wp, cout, residual=res)
    assert (zr.cpu() - (want + res.cpu())).abs().max().item() <= tol
    mask = torch.randint(0, 16, (n, h, w, cout // 4), generator=g, dtype=torch.uint8).to(dev)
    bits = torch.stack([(mask >> k) & 1 for k in range(4)], -1).reshape(n, h, w, cout).float()
    zm = ops.conv3x3_hw(x, wp, cout, residual=res, res_mask=mask)
    assert (zm.cpu() - (want + (res * bits).cpu())).abs().max().item() <= tol
    tr = _bn_params(cin, g, dev)
    act = ops.bn_apply_fwd(x, tr[0], tr[1], tr[2], tr[3], None, True)
    z2, em, m2, i2 = ops.conv3x3_hw(x, wp, cout, transform=tr, emit=True, stats=(1e-5, 0.1, rm2, rv2))
    assert torch.equal(em, act)                                                     # the same expression
    assert torch.equal(z2, ops.conv3x3_hw(act, wp, cout))
    if ops.conv3x3_hw_ok(n, h, w, cout, cin, f32=True):
        wpf, _ = ops.conv3x3_hw_pack(w32.reshape(-1), [(0, cin, cout, True)], f32=True)
        dx = ops.conv3x3_hw(res, wpf, cin)
        wantdx = F.conv_transpose2d(res.cpu().permute(0, 3, 1, 2), w32.cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
        assert (dx.cpu() - wantdx).abs().max().item() <= 2e-5 * max(1.0, wantdx.abs().max().item())


@pytest.mark.parametrize("shape", [(1025, 16, 16, 64, 64), (64, 64, 64, 64, 64), (128, 32, 32, 128, 128)])
def test_conv3x3_fw_inference_form(dev, shape):
    """Inference epilogue of the register-fed conv (BatchNorm scale folded into the packed filter, shift + residual + ReLU, output NHWC or
    position-major) against the halo c64 kernel / the implicit GEMM with the same folded BatchNorm; two 16 x 16 maps per tile with an odd
    number of maps."""
    from self_supervised import ops
    n, h, w, cin, cout = shape
    assert ops.conv3x3_fw_eval_ok(n, h, w, cin, cout)
    g = torch.Generator().manual_seed(n + h)
    x = torch.randn(n, h, w, cin, generator=g).to(dev)
    wt = (torch.randn(cout, 3, 3, cin, generator=g) / (9 * cin) ** 0.5).to(dev)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    res = torch.randn(n, h, w, cout, generator=g).to(dev)
    wp = ops.conv3x3_fw_pack_scaled(wt, sc)
    # torch's own conv as the yardstick (the HIP-against-HIP comparison alone would let a shared mistake through); the
    # largest case stays HIP-against-HIP only on the CPU's account (a 1025-image F.conv2d is fine, it is the 64 x 64 maps that cost)
    ref = None
    if n * h * w <= 1025 * 256:
        ref = F.conv2d(x.cpu().permute(0, 3, 1, 2), wt.cpu().permute(0, 3, 1, 2), None, 1, 1) * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1)
    for r_, relu in ((None, True), (res, True), (res, False)):
        want = ops.conv_fwd(x, wt, sc, sh, r_, relu, 1, 1)
        got = ops.conv3x3_fw_eval(x, wp, cout, sh, r_, relu)
        tol = 2e-5 * max(1.0, want.abs().max().item())
        assert (got - want).abs().max().item() <= tol
        if ref is not None:
            t = ref + (r_.cpu().permute(0, 3, 1, 2) if r_ is not None else 0.0)
            t = (t.relu() if relu else t).permute(0, 2, 3, 1)
            assert (got.cpu() - t).abs().max().item() <= 2e-5 * max(1.0, t.abs().max().item())
        got_pm = ops.conv3x3_fw_eval(x, wp, cout, sh, r_, relu, out_hwnc=True)
        assert torch.equal(got_pm.permute(2, 0, 1, 3), got)


@pytest.mark.parametrize("shape", [(3, 16, 16, 64, 64, 1), (2, 13, 21, 64, 128, 1), (5, 8, 8, 128, 64, 1), (3, 5, 7, 64, 64, 1),
                                   (40, 32, 32, 128, 128, 1), (9, 64, 64, 64, 64, 1), (3, 32, 32, 64, 128, 2), (2, 13, 21, 64, 64, 2),
                                   (5, 16, 16, 128, 256, 2), (40, 16, 16, 256, 512, 2), (3, 9, 9, 64, 64, 2)])
def test_wgrad_gather16_kernel(dev, shape):
    """csrc/wgrad16.hip (fragments gathered from [pixel][channel] tiles by 2-byte LDS reads) against fp32 math on the same half
    operands: stride 1 and 2, 4 x 16 / 8 x 8 / 2 x 16 / 4 x 8 tiles, ragged tiles, odd maps, several (co, ci) blocks and splits."""
    from self_supervised import ops
    n, h, w, cin, cout, s_ = shape
    g = torch.Generator().manual_seed(n * 5 + h + cin + s_)
    x = torch.randn(n, h, w, cin, generator=g).half()
    ho, wo = (h - 1) // s_ + 1, (w - 1) // s_ + 1
    dy = torch.randn(n, ho, wo, cout, generator=g).half()
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(False)
    wref = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(xr, wref, None, s_, 1).backward(dy.float().permute(0, 3, 1, 2))
    want = wref.grad.permute(0, 2, 3, 1).reshape(-1)                    # OHWI
    dw = torch.empty(cout * 9 * cin, device=dev)
    ops.conv_wgrad(dy.to(dev), x.to(dev), dw, 3, 3, s_, 1, bf16=2)
    assert (dw.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), shape


def test_elementwise_half_tensors(dev):
    """BatchNorm apply / backward reductions / backward apply, global average pooling both ways: half tensors == the fp32-tensor
    kernels on the same values, outputs rounded once, parameter gradients equal."""
    from self_supervised import ops
    for (n, h, w, c) in [(3, 8, 8, 64), (2, 5, 7, 128), (33, 4, 4, 512)]:
        g = torch.Generator().manual_seed(n + c)
        z = _r(torch.randn(n, h, w, c, generator=g)).to(dev)
        res = _r(torch.randn(n, h, w, c, generator=g)).to(dev)
        dy = _r(torch.randn(n, h, w, c, generator=g)).to(dev)
        mean, invstd, gamma, beta = _bn_params(c, g, dev)
        for r_, relu in ((None, True), (res, True), (res, False), (None, False)):
            y32 = ops.bn_apply_fwd(z, mean, invstd, gamma, beta, r_, relu)
            yh = ops.bn_apply_fwd(z.half(), mean, invstd, gamma, beta, None if r_ is None else r_.half(), relu)
            assert yh.dtype == torch.float16 and torch.equal(yh, y32.half())
        m1, i1 = ops.bn_stats(z, c, 1e-5, 0.1, None, None)
        m2, i2 = ops.bn_stats(z.half(), c, 1e-5, 0.1, None, None)
        assert torch.equal(m1, m2) and torch.equal(i1, i2)
        # backward with the mask from the stored activation (residual block) ...
        yact = ops.bn_apply_fwd(z, mean, invstd, gamma, beta, res, True).half().float()
        db1, dg1, db2, dg2 = (torch.empty(c, device=dev) for _ in range(4))
        ops.bn_bwd_reduce(dy, yact, z, mean, invstd, db1, dg1, c)
        ops.bn_bwd_reduce(dy.half(), yact.half(), z.half(), mean, invstd, db2, dg2, c)
        assert torch.equal(db1, db2) and torch.equal(dg1, dg2)
        dz32, dr32 = ops.bn_apply_bwd(dy, yact, z, mean, invstd, gamma, db1, dg1, True)
        dzh, drh = ops.bn_apply_bwd(dy.half(), yact.half(), z.half(), mean, invstd, gamma, db1, dg1, True)
        assert torch.equal(dzh, dz32.half()) and torch.equal(drh, dr32.half())
        # ... and with the mask recomputed from z
        dz32 = ops.bn_bwd_zmask(dy, z, mean, invstd, gamma, beta, db1, dg1)
        dzh = ops.bn_bwd_zmask(dy.half(), z.half(), mean, invstd, gamma, beta, db2, dg2)
        assert torch.equal(db1, db2) and torch.equal(dg1, dg2) and torch.equal(dzh, dz32.half())
        # global average pooling
        if c % 64 == 0:
            p1, p2 = torch.zeros(n, c + 64, device=dev), torch.zeros(n, c + 64, device=dev)
            ops.gap_fwd(z, p1, 64)
            ops.gap_fwd(z.half(), p2, 64)
            assert (p1 - p2).abs().max().item() <= 1e-6 * max(1.0, p1.abs().max().item())
        dp = torch.randn(n, c + 64, generator=g).to(dev)
        d1, d2 = dy.clone(), dy.half()
        ops.gap_bwd(dp, d1, 64, True)
        ops.gap_bwd(dp, d2, 64, True)
        assert torch.equal(d2, d1.half())
        ops.gap_bwd(dp, d1, 64, False)
        ops.gap_bwd(dp, d2, 64, False)
        assert torch.equal(d2, d1.half())


def test_stem_half_tensors(dev):
    """conv1 (fp16 operands) with z stored as halves, BatchNorm + ReLU + max-pool over it, their fused backward and the stem weight
    gradient from a half dz."""
    from self_supervised import ops
    from oracle import weights as ow
    for (b, hw) in [(3, 64), (2, 96)]:
        g = torch.Generator().manual_seed(b * 7 + hw)
        img = ow.synthetic_images(b, hw, seed=40 + b).to(dev)
        w = (torch.randn(64, 3, 7, 7, generator=g) / 12.0).to(dev)
        rm1, rv1 = torch.zeros(64, device=dev), torch.ones(64, device=dev)
        rm2, rv2 = rm1.clone(), rv1.clone()
        z32, m32, i32 = ops.stem_fwd_stats16(img, w, 1e-5, 0.1, rm1, rv1, 2)
        zh, mh, ih = ops.stem_fwd_stats16(img, w, 1e-5, 0.1, rm2, rv2, 2, out_half=True)
        assert zh.dtype == torch.float16 and torch.equal(zh, z32.half())
        zc = zh.double().reshape(-1, 64)
        assert (mh.double() - zc.mean(0)).abs().max().item() < 1e-5 * max(1.0, zc.abs().max().item())
        assert ((ih.double() - (zc.var(0, unbiased=False) + 1e-5).rsqrt()).abs() / ih.double()).max().item() < 1e-5
        gamma, beta = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
        # pooling compares the BatchNorm outputs AS HALVES (autocast's BatchNorm output is a half tensor): the fp32 pooling kernel over
        # that rounded activation is the statement
        zf = zh.float()
        act = ops.bn_apply_fwd(zf, mh, ih, gamma, beta, None, True).half().float()
        want, widx = ops.maxpool3x3s2_fwd_idx(act)
        got, gidx = ops.bn_relu_maxpool_fwd(zh, mh, ih, gamma, beta)
        assert got.dtype == torch.float16 and torch.equal(got, want.half()) and torch.equal(gidx, widx)
        # fused backward: pooled gradient -> dz, dbeta, dgamma
        dpool = _r(torch.randn(got.shape, generator=g)).to(dev)
        db1, dg1, db2, dg2 = (torch.empty(64, device=dev) for _ in range(4))
        dz32 = ops.pool_bn_relu_bwd(gidx, dpool, zf, mh, ih, gamma, beta, db1, dg1)
        dzh = ops.pool_bn_relu_bwd(gidx, dpool.half(), zh, mh, ih, gamma, beta, db2, dg2)
        assert torch.equal(db1, db2) and torch.equal(dg1, dg2) and torch.equal(dzh, dz32.half())
        # conv1's weight gradient from the half dz on fp16 operands (the image rounded while staged, as autocast feeds it): the exact
        # fp32 kernel over the same rounded values gives the same products in another summation order
        dw1, dw2 = torch.empty(64 * 147, device=dev), torch.empty(64 * 147, device=dev)
        ops.stem_wgrad(img.half().float(), dzh.float(), dw1)
        ops.stem_wgrad(img, dzh, dw2)
        assert (dw1 - dw2).abs().max().item() <= 2e-5 * dw1.abs().max().item()


def test_wrong_extents_raise(dev):
    """include/ssad.h: "return non-zero ... Never aborts".  Every entry point that reads a second tensor over extents derived from
    the first one's takes that buffer's element count; a count that does not fit is an error return (HipExtensionError through the
    binding) -- not an out-of-bounds read (round 4: a half-size dz reached ssad_stem_wgrad).  One case per guarded entry, through the C
    ABI itself; the tensor wrappers refuse the same shapes before they get that far."""
    from self_supervised import _hip, ops
    from self_supervised._hip import HipExtensionError
    lib, st = _hip.lib(), _hip.stream()
    f = lambda *shape: torch.zeros(shape, device=dev)
    h_ = lambda *shape: torch.zeros(shape, device=dev, dtype=torch.float16)
    P = _hip.ptr
    img, dw = f(2, 3, 64, 64), f(64 * 147)
    ws = f(int(lib.ssad_stem_wgrad_workspace(2, 64, 64)))
    dz_small = f(2, 16, 16, 64)                       # belongs to 32 x 32 images
    with pytest.raises(HipExtensionError, match="dz does not hold"):
        _hip.check(lib.ssad_stem_wgrad(P(img), P(dz_small), P(dw), 2, 64, 64, dz_small.numel(), 0, 0, P(ws), st))
    with pytest.raises(HipExtensionError, match="dz does not hold"):
        _hip.check(lib.ssad_stem_wgrad_h(P(img), h_(2, 16, 16, 64).data_ptr(), P(dw), 2, 64, 64, dz_small.numel(), 0, 0, P(ws), st))
    with pytest.raises(AssertionError):
        ops.stem_wgrad(img, dz_small, dw)
    # pooled gradient / slots of the wrong map size
    z, dpool = f(2, 32, 32, 64), f(2, 8, 8, 64)
    idx = torch.zeros(2, 8, 8, 64, device=dev, dtype=torch.uint8)
    v = [f(64) for _ in range(6)]
    wsd = torch.zeros(int(lib.ssad_colreduce_workspace(2 * 32 * 32, 64)), device=dev, dtype=torch.float64)
    with pytest.raises(HipExtensionError, match="dpool"):
        _hip.check(lib.ssad_pool_bn_relu_bwd(idx.data_ptr(), P(dpool), P(z), P(v[0]), P(v[1]), P(v[2]), P(v[3]), P(v[4]), P(v[5]),
                                             P(torch.empty_like(z)), 2, 32, 32, 64, dpool.numel(), wsd.data_ptr(), st))
    with pytest.raises(HipExtensionError, match="dpool"):
        _hip.check(lib.ssad_pool_bn_relu_bwd_h(idx.data_ptr(), dpool.half().data_ptr(), z.half().data_ptr(), P(v[0]), P(v[1]), P(v[2]),
                                               P(v[3]), P(v[4]), P(v[5]), torch.empty_like(z).half().data_ptr(), 2, 32, 32, 64,
                                               dpool.numel(), wsd.data_ptr(), st))
    with pytest.raises(AssertionError):
        ops.pool_bn_relu_bwd(idx, dpool, z, v[0], v[1], v[2], v[3], v[4], v[5])
    with pytest.raises(HipExtensionError, match="dy"):
        _hip.check(lib.ssad_maxpool3x3s2_bwd_idx(idx.data_ptr(), P(dpool), P(torch.empty_like(z)), 2, 32, 32, 64, dpool.numel(), st))
    with pytest.raises(HipExtensionError, match="dy"):
        _hip.check(lib.ssad_maxpool3x3s2_bwd(P(z), P(dpool), P(torch.empty_like(z)), 2, 32, 32, 64, dpool.numel(), st))
    with pytest.raises(AssertionError):
        ops.maxpool3x3s2_bwd_idx(idx, dpool, z.shape)
    # weight gradients: dy of the wrong map size for x and the filter geometry
    x, dy_bad = f(2, 8, 8, 64), f(2, 4, 4, 64)        # a stride-1 3 x 3 / pad 1 conv over 8 x 8 gives 8 x 8
    slab = f(64, 64, 9 * 64)
    for fn in (lib.ssad_conv_wgrad, lib.ssad_conv_wgrad_bf16, lib.ssad_conv_wgrad_f16, lib.ssad_conv_wgrad_x3, lib.ssad_conv_wgrad_x6):
        with pytest.raises(HipExtensionError, match="dy does not hold"):
            _hip.check(fn(P(dy_bad), P(x), P(slab), 1, 2, 8, 8, 64, 64, 3, 3, 1, 1, dy_bad.numel(), st))
    with pytest.raises(HipExtensionError, match="dy does not hold"):
        _hip.check(lib.ssad_conv_wgrad_f16_h(dy_bad.half().data_ptr(), x.half().data_ptr(), P(slab), 1, 2, 8, 8, 64, 64, 3, 3, 1, 1,
                                             dy_bad.numel(), st))
    sp = lib.ssad_wgrad3x3_halo_splits(2, 8, 8, 64, 64)
    with pytest.raises(HipExtensionError, match="dz does not hold"):
        _hip.check(lib.ssad_conv_wgrad3x3_halo(P(dy_bad), P(x), P(slab), sp, 2, 8, 8, 64, 64, dy_bad.numel(), st))
    with pytest.raises(HipExtensionError, match="dz does not hold"):
        _hip.check(lib.ssad_conv_wgrad3x3s2_halo(P(f(2, 2, 2, 64)), P(x), P(slab), lib.ssad_wgrad3x3_halo_splits(2, 4, 4, 64, 64), 2, 4, 4,
                                                 8, 8, 64, 64, 2 * 2 * 2 * 64, st))
    sp16 = lib.ssad_wgrad3x3_halo16_splits(2, 8, 8, 64, 64)
    with pytest.raises(HipExtensionError, match="dz does not hold"):
        _hip.check(lib.ssad_conv_wgrad3x3_halo16(P(dy_bad), P(x), P(slab), sp16, 2, 8, 8, 64, 64, 1, dy_bad.numel(), st))
    with pytest.raises(HipExtensionError, match="dz does not hold"):
        _hip.check(lib.ssad_conv_wgrad3x3_halo16_h(dy_bad.half().data_ptr(), x.half().data_ptr(), P(slab), sp16, 2, 8, 8, 64, 64,
                                                   dy_bad.numel(), st))
    with pytest.raises(AssertionError):
        ops.conv_wgrad(dy_bad, x, f(64 * 9 * 64), 3, 3, 1, 1)
    # tensors that must share one shape: the wrappers say so
    with pytest.raises(AssertionError):
        ops.bn_apply_bwd(f(2, 8, 8, 64), None, f(2, 4, 4, 64), v[0], v[1], v[2], v[3], v[4], False)
    with pytest.raises(AssertionError):
        ops.bn_apply_fwd(f(2, 8, 8, 64), v[0], v[1], v[2], v[3], f(2, 4, 4, 64), True)
    with pytest.raises(AssertionError):
        ops.conv_dgrad(f(2, 8, 8, 64), f(64, 3, 3, 64), (2, 8, 8, 64), 1, 1, residual=f(2, 4, 4, 64))
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------
# round 6: nibble masks of the residual blocks over half tensors
# ---------------------------------------------------------------------------------------------
def test_half_nibble_masks(dev):
    """ssad_bn_apply_fwd_mask_h / _bwd_reduce_mask_h / _apply_bwd_mask_h and the residual mask of ssad_conv3x3_hw: the mask is the sign
    pattern of the STORED half activation (two bytes per 8 channels, nibble k = channel quad k as in the fp32 form), and every consumer
    gives bit for bit what it gives when the activation / the pre-masked gradient is handed to it instead."""
    from self_supervised import ops
    g = torch.Generator().manual_seed(21)
    n, h, w, c = 208, 16, 32, 64                      # 208 tiles of 16 x 32 x 64: inside ssad_conv3x3_hw_ok
    mean, invstd, gamma, beta = _bn_params(c, g, dev)
    z = torch.randn(n, h, w, c, generator=g).half().to(dev)
    res = torch.randn(n, h, w, c, generator=g).half().to(dev)
    y_ref = ops.bn_apply_fwd(z, mean, invstd, gamma, beta, res, True)
    y, mask = ops.bn_apply_fwd_mask(z, mean, invstd, gamma, beta, res, True)
    assert y.dtype == torch.float16 and torch.equal(y, y_ref)
    bits = mask.view(n, h, w, c // 4)
    want = (y > 0).view(n, h, w, c // 4, 4).to(torch.uint8)
    want = want[..., 0] | (want[..., 1] << 1) | (want[..., 2] << 2) | (want[..., 3] << 3)
    assert torch.equal(bits, want)
    # backward through the BatchNorm: mask == the saved activation
    dy = torch.randn(n, h, w, c, generator=g).half().to(dev)
    db0, dg0 = torch.empty(c, device=dev), torch.empty(c, device=dev)
    ops.bn_bwd_reduce(dy, y, z, mean, invstd, db0, dg0, c)
    dz0, dres0 = ops.bn_apply_bwd(dy, y, z, mean, invstd, gamma, db0, dg0, True)
    db1, dg1 = torch.empty(c, device=dev), torch.empty(c, device=dev)
    dz1 = ops.bn_bwd_mask(dy, mask, z, mean, invstd, gamma, db1, dg1)
    assert torch.equal(db0, db1) and torch.equal(dg0, dg1) and torch.equal(dz0, dz1)
    # the identity-branch gradient as (dy, mask) into the register-fed conv == the materialised one
    assert ops.conv3x3_hw_ok(n, h, w, c, c)
    wt = (torch.randn(c, 3, 3, c, generator=g) / 24).to(dev)
    packed, _ = ops.conv3x3_hw_pack(wt.flatten(), [(0, c, c, False)])
    x = torch.randn(n, h, w, c, generator=g).half().to(dev)
    a = ops.conv3x3_hw(x, packed, c, residual=dres0)
    b = ops.conv3x3_hw(x, packed, c, residual=dy, res_mask=mask)
    assert torch.equal(a, b)


def test_half_masks_leave_the_step_bit_identical(dev):
    """A precision-16 step at 32 x 256 x 256 (layer1 on the register-fed conv with masks, the smaller layers on csrc/conv16.hip without:
    both branches of TrainEngine.use_relu_mask16) with SSAD_MASK16 on and off: same g everywhere, hence the same bits in every
    parameter, momentum and running statistic after three steps."""
    from oracle import weights as ow
    from self_supervised import training
    from self_supervised.models import PeraNet
    sd = ow.seeded_state_dict(0)
    x, y = ow.synthetic_images(32, 256, seed=31).to(dev), ow.synthetic_labels(32, seed=32).to(dev)
    states = []
    for on in (False, True):
        m = PeraNet(); m.load_state_dict(sd); m.to(dev).train(); m.unfreeze()
        step = training.DataParallelStep(m, lr=0.01, world_size=1, precision=16, graph=False)
        step.eng.sw_mask16 = on
        for _ in range(3):
            step.step(x, y)
        torch.cuda.synchronize()
        assert step.eng.h16
        states.append(torch.cat([step.eng.arena.p, step.eng.arena.m] + [b.detach().flatten().float() for b in m.buffers()]).clone())
    assert torch.equal(states[0], states[1])
