"""GPU parity of the training step: HIP forward(train)/backward/SGD against torch-CPU autograd on the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rel_err(got, want, floor=1e-12):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    return (got - want).abs().max().item() / max(want.abs().max().item(), floor)


def grad_floor(ref):
    """Gradients that are analytically zero (a BN bias feeding another train-mode BN) are rounding noise on both
    sides: compare every tensor against max(|its own reference|, 1e-4 * the largest gradient in the model)."""
    return 1e-4 * max(p.grad.abs().max().item() for p in ref.parameters() if p.grad is not None)


def _pair(sd, dev):
    from self_supervised.models import PeraNet
    from oracle.peranet import OraclePeraNet
    ref = OraclePeraNet(); ref.load_state_dict(sd); ref.train()
    m = PeraNet(); m.load_state_dict(sd); m.to(dev).train()
    return ref, m


def test_wgrad_dgrad_kernels(dev):
    from self_supervised import ops
    for (n, h, w, cin, cout, k, s, p) in [(3, 8, 8, 64, 64, 3, 1, 1), (2, 9, 9, 64, 128, 3, 2, 1), (2, 8, 8, 64, 128, 1, 2, 0),
                                           (5, 2, 2, 128, 256, 3, 1, 1), (300, 1, 1, 896, 512, 1, 1, 0)]:
        g = torch.Generator().manual_seed(n * 7 + k)
        x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
        wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).requires_grad_()
        y = F.conv2d(x, wt, None, s, p)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        xd, dyd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev), dy.permute(0, 2, 3, 1).contiguous().to(dev)
        w_ohwi = ops.repack_oihw_to_ohwi(wt.detach().to(dev))
        dx = ops.conv_dgrad(dyd, ops.flip_transpose_weight(w_ohwi), xd.shape, s, p)
        assert rel_err(dx.permute(0, 3, 1, 2), x.grad) < 2e-5
        dw = torch.empty(cout * k * k * cin, device=dev)
        ops.conv_wgrad(dyd, xd, dw, k, k, s, p)
        assert rel_err(dw.view(cout, k, k, cin).permute(0, 3, 1, 2), wt.grad) < 2e-5
        dw2 = torch.empty(cout * k * k * cin, device=dev)
        ops.conv_wgrad(dyd, xd, dw2, k, k, s, p, to_oihw=True)
        assert torch.equal(dw2.view(cout, cin, k, k), dw.view(cout, k, k, cin).permute(0, 3, 1, 2))


@pytest.mark.parametrize("m", [1, 5, 32, 33, 256, 512])
def test_linear_small_kernels(dev, m):
    """Linear layers over a training batch's rows (csrc/linear_small.hip): forward with the fused epilogue, forward + train-mode
    BatchNorm1d statistics, input gradient (any contraction that is a multiple of 4: the 4-class classifier) and weight gradient,
    against float64 torch; ragged row and column tiles; the same layer one row above the range (implicit-GEMM kernel) agrees."""
    from self_supervised import _hip, ops
    assert _hip.lib().ssad_linear_small_max_rows() == 512
    for (k, n) in [(512, 512), (896, 512), (512, 4), (512, 70), (36, 44)]:
        g = torch.Generator().manual_seed(m * 131 + k + n)
        x = torch.randn(m, k, generator=g, dtype=torch.float64)
        wt = torch.randn(n, k, generator=g, dtype=torch.float64) / k ** 0.5
        sc, sh = torch.rand(n, generator=g, dtype=torch.float64) + 0.5, torch.randn(n, generator=g, dtype=torch.float64)
        r = torch.randn(m, n, generator=g, dtype=torch.float64)
        z = x @ wt.T
        xd, wd = x.float().view(m, 1, 1, k).to(dev), wt.float().view(n, 1, 1, k).to(dev)
        got = ops.conv_fwd(xd, wd, sc.float().to(dev), sh.float().to(dev), r.float().view(m, 1, 1, n).to(dev), True, 1, 0)
        assert rel_err(got.view(m, n), (z * sc + sh + r).relu()) < 2e-6
        got = ops.conv_fwd(xd, wd, None, sh.float().to(dev), None, False, 1, 0)          # bias only (the classifier)
        assert rel_err(got.view(m, n), z + sh) < 2e-6
        if n % 4 == 0 and m > 1:
            rm, rv = torch.zeros(n, device=dev), torch.ones(n, device=dev)
            zz, mean, invstd = ops.conv_fwd_stats(xd, wd, 1e-5, 0.1, rm, rv, 1, 0)
            zf = zz.view(m, n).double().cpu()
            assert rel_err(zz.view(m, n), z) < 2e-6
            assert rel_err(mean, zf.mean(0)) < 1e-6 and rel_err(invstd, (zf.var(0, unbiased=False) + 1e-5).rsqrt()) < 1e-5
            assert rel_err(rm, 0.1 * zf.mean(0)) < 1e-6 and rel_err(rv, 0.9 + 0.1 * zf.var(0, unbiased=True)) < 1e-5
        # input gradient: dx[m][k] = dz[m][n] . w[n][k]; the contraction is n (4, 44, 70 are not multiples of 32)
        dz = torch.randn(m, n, generator=g, dtype=torch.float64)
        if n % 4 == 0:
            wf = ops.flip_transpose_weight(wd)
            dx = ops.conv_dgrad(dz.float().view(m, 1, 1, n).to(dev), wf, (m, 1, 1, k), 1, 0, residual=xd)
            assert rel_err(dx.view(m, k), dz @ wt + x) < 2e-6
        dw = torch.full((n * k,), 0.5, device=dev)
        ops.conv_wgrad(dz.float().view(m, 1, 1, n).to(dev), xd, dw, 1, 1, 1, 0, accumulate=True)
        assert rel_err(dw.view(n, k), dz.T @ x + 0.5) < 2e-6
        ops.conv_wgrad(dz.float().view(m, 1, 1, n).to(dev), xd, dw, 1, 1, 1, 0)
        assert rel_err(dw.view(n, k), dz.T @ x) < 2e-6
    if m == 512:        # one row more: the implicit-GEMM kernel takes over; both are fp32 contractions of the same numbers
        g = torch.Generator().manual_seed(5)
        x, wt = torch.randn(513, 512, generator=g), torch.randn(512, 512, generator=g) / 512 ** 0.5
        big = ops.conv_fwd(x.view(513, 1, 1, 512).to(dev), wt.view(512, 1, 1, 512).to(dev), None, None, None, False, 1, 0)
        small = ops.conv_fwd(x[:512].view(512, 1, 1, 512).to(dev), wt.view(512, 1, 1, 512).to(dev), None, None, None, False, 1, 0)
        assert rel_err(small.view(512, 512), big.view(513, 512)[:512]) < 2e-6


@pytest.mark.parametrize("rows", [8, 32, 45, 256, 512])
def test_bn_small_kernels(dev, rows):
    """One-launch BatchNorm over a training batch's rows (csrc/train.hip, bn_small_*): forward (statistics, running statistics,
    apply, ReLU) and backward (dbeta, dgamma, dz, bias gradient of the Linear in front) against float64 autograd, and against
    the three-launch path they replace."""
    from self_supervised import ops
    assert ops.bn_small_ok(rows, 512) and not ops.bn_small_ok(513, 512) and not ops.bn_small_ok(rows, 48)
    for c, relu in [(512, True), (512, False), (64, True)]:
        g = torch.Generator().manual_seed(rows + c + relu)
        z = (torch.randn(rows, c, generator=g, dtype=torch.float64) * 1.7 + 0.3).requires_grad_()
        bn = torch.nn.BatchNorm1d(c).double()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(c, generator=g, dtype=torch.float64) + 0.5)
            bn.bias.copy_(torch.randn(c, generator=g, dtype=torch.float64) * 0.3)
        y = bn(z)
        y = y.relu() if relu else y
        dy = torch.randn(rows, c, generator=g, dtype=torch.float64)
        y.backward(dy)
        f = lambda t: t.detach().float().to(dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        yy, mean, invstd = ops.bn_small_fwd(f(z), f(bn.weight), f(bn.bias), bn.eps, 0.1, rm, rv, relu)
        assert rel_err(yy, y) < 1e-5 and rel_err(rm, bn.running_mean) < 1e-6 and rel_err(rv, bn.running_var) < 1e-6     # mean kept in float32: two close rows cancel
        db, dg, dbias = torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(c, device=dev)
        dz = ops.bn_small_bwd(f(dy), f(z), mean, invstd, f(bn.weight), f(bn.bias) if relu else None, db, dg, dbias)
        tol = 2e-6 if rows >= 32 else 2e-5        # few rows: xhat carries the rounding of the float32 mean and invstd
        assert rel_err(db, bn.bias.grad) < tol and rel_err(dg, bn.weight.grad) < tol
        assert rel_err(dz, z.grad, floor=1e-3) < 5 * tol
        assert dbias.abs().max().item() < 1e-4 * max(1.0, dz.abs().max().item()) * rows    # analytically zero: sum of dz over the batch
        assert rel_err(dbias, dz.double().sum(0), floor=1e-3) < 1e-3
        # the general path: col_reduce + finalize + apply
        m2, i2 = ops.bn_stats(f(z), c, bn.eps, 0.1, None, None)
        assert rel_err(mean, m2) < 1e-6 and rel_err(invstd, i2) < 1e-5
        y2 = ops.bn_apply_fwd(f(z), m2, i2, f(bn.weight), f(bn.bias), None, relu)
        assert rel_err(yy, y2) < 1e-5


def test_bn_pool_kernels(dev):
    from self_supervised import ops
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(6, 64, 10, 10, generator=g) * 2 + 0.5).requires_grad_()
    bn = torch.nn.BatchNorm2d(64)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(64, generator=g) + 0.5); bn.bias.copy_(torch.randn(64, generator=g))
    res = torch.randn(6, 64, 10, 10, generator=g)
    y = F.relu(bn(x) + res)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    mean, invstd = ops.bn_stats(nh(x), 64, bn.eps, 0.1, rm, rv)
    assert rel_err(rm, bn.running_mean) < 1e-5 and rel_err(rv, bn.running_var) < 1e-5
    yy = ops.bn_apply_fwd(nh(x), mean, invstd, bn.weight.detach().to(dev), bn.bias.detach().to(dev), nh(res), True)
    assert rel_err(yy, nh(y)) < 1e-5
    db, dg = torch.empty(64, device=dev), torch.empty(64, device=dev)
    ops.bn_bwd_reduce(nh(dy), yy, nh(x), mean, invstd, db, dg, 64)
    assert rel_err(db, bn.bias.grad) < 1e-5 and rel_err(dg, bn.weight.grad) < 1e-5
    dz, dres = ops.bn_apply_bwd(nh(dy), yy, nh(x), mean, invstd, bn.weight.detach().to(dev), db, dg, True)
    assert rel_err(dz, nh(x.grad)) < 2e-5
    # max-pool backward incl. ties (relu zeros) and gap backward
    a = F.relu(torch.randn(3, 64, 9, 11, generator=g)).requires_grad_()
    p = F.max_pool2d(a, 3, 2, 1)
    dp = torch.randn(p.shape, generator=g)
    p.backward(dp)
    da = ops.maxpool3x3s2_bwd(nh(a), nh(dp))
    assert torch.equal(da.cpu(), nh(a.grad).cpu())
    pooled, idx = ops.maxpool3x3s2_fwd_idx(nh(a))
    assert torch.equal(pooled.cpu(), nh(p).cpu())
    assert torch.equal(ops.maxpool3x3s2_bwd_idx(idx, nh(dp), nh(a).shape).cpu(), nh(a.grad).cpu())


@pytest.mark.parametrize("half", [False, True])
def test_stem_reduction_over_pooled_tensors(dev, half):
    """BatchNorm + ReLU + max-pool of the stem, backward: the reduction over (dpool, z of each window's winner) -- a quarter of the rows,
    no pass over z -- gives the parameter gradients and the dz of the two-pass form over z (same sums in another order), and both are
    autograd's; odd map sizes, ties on ReLU zeros, fp32 and half tensors."""
    from self_supervised import ops
    g = torch.Generator().manual_seed(5)
    z = (torch.randn(5, 64, 21, 18, generator=g) * 2 - 0.3).requires_grad_()
    bn = torch.nn.BatchNorm2d(64)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(64, generator=g) + 0.5); bn.bias.copy_(torch.randn(64, generator=g) * 0.3)
    p = F.max_pool2d(F.relu(bn(z)), 3, 2, 1)
    dp = torch.randn(p.shape, generator=g)
    p.backward(dp)
    nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    zz, dpp = nh(z), nh(dp)
    if half:
        zz, dpp = zz.half(), dpp.half()
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    mean, invstd = ops.bn_stats(zz, 64, bn.eps, 0.1, rm, rv)
    ga, be = bn.weight.detach().to(dev), bn.bias.detach().to(dev)
    pooled, idx = ops.bn_relu_maxpool_fwd(zz, mean, invstd, ga, be)
    pooled2, idx2, zwin = ops.bn_relu_maxpool_fwd(zz, mean, invstd, ga, be, winners=True)
    assert torch.equal(pooled, pooled2) and torch.equal(idx, idx2)
    # the winner's raw value: relu(bn(zwin)) is the pooled value (up to the kernel's fused multiply-add / the half rounding)
    act = torch.relu((zwin.float() - mean) * invstd * ga + be)
    assert (act - pooled.float()).abs().max().item() <= (2e-3 if half else 2e-6) * max(1.0, pooled.float().abs().max().item())
    db1, dg1, db2, dg2 = (torch.empty(64, device=dev) for _ in range(4))
    dz1 = ops.pool_bn_relu_bwd(idx, dpp, zz, mean, invstd, ga, be, db1, dg1)
    dz2 = ops.pool_bn_relu_bwd(idx, dpp, zz, mean, invstd, ga, be, db2, dg2, zwin=zwin)
    assert rel_err(db2, db1) < 1e-6 and rel_err(dg2, dg1) < 1e-6
    assert rel_err(dz2, dz1) < (2e-3 if half else 1e-6)
    if not half:
        assert rel_err(db2, bn.bias.grad) < 1e-5 and rel_err(dg2, bn.weight.grad) < 1e-5 and rel_err(dz2, nh(z.grad)) < 2e-5


def test_stem_wgrad_kernel(dev):
    """conv1 weight gradient straight from the NCHW image vs torch autograd (incl. ragged tiles and the <64 px resize)."""
    from self_supervised import ops
    for (b, h, w) in [(3, 64, 64), (2, 70, 90), (5, 32, 32), (2, 256, 256)]:
        g = torch.Generator().manual_seed(b * h)
        img = torch.randn(b, 3, h, w, generator=g)
        wt = (torch.randn(64, 3, 7, 7, generator=g) / 12).requires_grad_()
        src = F.interpolate(img, (64, 64), mode="nearest") if (h < 64 or w < 64) else img
        z = F.conv2d(src, wt, None, 2, 3)
        dz = torch.randn(z.shape, generator=g)
        z.backward(dz)
        dzd = dz.permute(0, 2, 3, 1).contiguous().to(dev)
        dw = torch.empty(64 * 147, device=dev)
        ops.stem_wgrad(img.to(dev), dzd, dw)
        assert rel_err(dw.view(64, 7, 7, 3).permute(0, 3, 1, 2), wt.grad) < 2e-5
        dw2 = torch.ones(64 * 147, device=dev)
        ops.stem_wgrad(img.to(dev), dzd, dw2, to_oihw=True, accumulate=True)
        assert rel_err(dw2.view(64, 3, 7, 7) - 1, wt.grad) < 2e-5
        dw3 = torch.empty(64 * 147, device=dev)
        ops.stem_wgrad(img.to(dev), dzd, dw3)
        assert torch.equal(dw, dw3)            # deterministic


def test_fused_stem_pool_kernels(dev):
    """BN + ReLU + max-pool forward and pool + ReLU + BN backward in one pass each == the unfused kernel chain."""
    from self_supervised import ops
    for (n, h, w) in [(3, 16, 16), (2, 9, 13)]:
        g = torch.Generator().manual_seed(n * h)
        z = (torch.randn(n, h, w, 64, generator=g) * 1.5 + 0.2).to(dev)
        gamma, beta = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
        mean, invstd = ops.bn_stats(z, 64, 1e-5, 0.1, None, None)
        y = ops.bn_apply_fwd(z, mean, invstd, gamma, beta, None, True)
        p1, i1 = ops.maxpool3x3s2_fwd_idx(y)
        p2, i2 = ops.bn_relu_maxpool_fwd(z, mean, invstd, gamma, beta)
        assert torch.equal(p1, p2) and torch.equal(i1, i2)
        dpool = torch.randn(p1.shape, generator=g).to(dev)
        da = ops.maxpool3x3s2_bwd_idx(i1, dpool, z.shape)
        db1, dg1 = torch.empty(64, device=dev), torch.empty(64, device=dev)
        dz1 = ops.bn_bwd_zmask(da, z, mean, invstd, gamma, beta, db1, dg1)
        db2, dg2 = torch.empty(64, device=dev), torch.empty(64, device=dev)
        dz2 = ops.pool_bn_relu_bwd(i1, dpool, z, mean, invstd, gamma, beta, db2, dg2)
        assert rel_err(db2, db1) < 1e-6 and rel_err(dg2, dg1) < 1e-6      # fp64 sums, different block partition
        assert rel_err(dz2, dz1) < 1e-6


def test_fused_stats_and_zmask_kernels(dev):
    """Conv epilogue statistics == separate bn_stats (bit-for-bit inputs, fp64 sums); BN backward with the ReLU mask
    recomputed from z == the same kernels reading the saved activation."""
    from self_supervised import ops
    for (n, h, w, cin, cout, k, s, p, bf) in [(5, 9, 9, 64, 64, 3, 1, 1, False), (3, 8, 8, 64, 128, 3, 2, 1, False),
                                               (130, 2, 2, 256, 512, 3, 1, 1, False), (4, 16, 16, 160, 64, 1, 1, 0, False),
                                               (3, 8, 8, 64, 128, 3, 1, 1, True)]:
        g = torch.Generator().manual_seed(n + cout)
        x = (torch.randn(n, h, w, cin, generator=g) + 0.3).to(dev)
        wt = (torch.randn(cout, k, k, cin, generator=g) / (cin * k * k) ** 0.5).to(dev)
        rm1, rv1 = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
        rm2, rv2 = rm1.clone(), rv1.clone()
        z1 = ops.conv_fwd(x, wt, None, None, None, False, s, p, bf)
        m1, i1 = ops.bn_stats(z1, cout, 1e-5, 0.1, rm1, rv1)
        z2, m2, i2 = ops.conv_fwd_stats(x, wt, 1e-5, 0.1, rm2, rv2, s, p, bf)
        assert torch.equal(z1, z2)
        # both are fp64 sums of the same fp32 values, only the summation tree differs
        assert rel_err(m2, m1) < 1e-6 and rel_err(i2, i1) < 1e-6
        assert rel_err(rm2, rm1) < 1e-6 and rel_err(rv2, rv1) < 1e-6
        zd = z1.double()
        assert rel_err(m2, zd.mean((0, 1, 2))) < 1e-6
        assert rel_err(i2, (zd.var((0, 1, 2), unbiased=False) + 1e-5).rsqrt()) < 1e-6
        # mask-from-z backward
        gamma, beta = torch.rand(cout, generator=g).to(dev) + 0.5, torch.randn(cout, generator=g).to(dev) * 0.3
        y = ops.bn_apply_fwd(z1, m1, i1, gamma, beta, None, True)
        dy = torch.randn(z1.shape, generator=g).to(dev)
        db1, dg1 = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
        ops.bn_bwd_reduce(dy, y, z1, m1, i1, db1, dg1, cout)
        dz1, _ = ops.bn_apply_bwd(dy, y, z1, m1, i1, gamma, db1, dg1, False)
        db2, dg2 = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
        dz2 = ops.bn_bwd_zmask(dy, z1, m1, i1, gamma, beta, db2, dg2)
        assert torch.equal(db1, db2) and torch.equal(dg1, dg2) and torch.equal(dz1, dz2)


def test_training_step_matches_autograd(dev, golden, seeded_sd):
    from self_supervised import training
    from oracle import weights as ow
    from oracle.peranet import train_step, make_optimizer
    g = golden("train_step")
    ref, m = _pair(seeded_sd, dev)
    x, y = ow.synthetic_images(8, 64, seed=55), ow.synthetic_labels(8, seed=56)
    loss_ref, acc_ref, out_ref = train_step(ref, x, y)
    loss_ref.backward()
    m.unfreeze()
    step = training.DataParallelStep(m, lr=0.03, world_size=1)
    eng = step.eng
    logits, emb = eng.forward(x.to(dev))
    assert rel_err(logits, out_ref["classifier"]) < 1e-4 and rel_err(emb, out_ref["latent_space"]) < 1e-4
    dlogits = torch.empty_like(logits)
    from self_supervised import ops
    la = ops.softmax_ce(logits, y.to(dev), dlogits, 1.0 / 8)
    np.testing.assert_allclose(la[0].item(), float(g["loss"]), rtol=1e-5)
    np.testing.assert_allclose(la[1].item(), acc_ref.item(), rtol=1e-6)
    eng.backward(dlogits)
    ref_params = dict(ref.named_parameters())
    worst = 0.0
    for name, p in m.named_parameters():
        e = rel_err(p.grad, ref_params[name].grad, grad_floor(ref))
        worst = max(worst, e)
        assert e < 1e-3, f"{name}: grad rel err {e:.3e}"
    for n, want in zip(g["grad_names"], g["grad_norms"]):
        got = dict(m.named_parameters())[str(n)].grad.double().norm().item()
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6)   # analytically-zero grads are rounding noise
    # fp64 truth: the HIP gradients must be as close to it as torch-CPU fp32 autograd is (same rounding class)
    import copy
    ref64 = copy.deepcopy(ref).double()
    ref64.zero_grad()
    l64, _, _ = train_step(ref64, x.double(), y)
    l64.backward()
    p64 = dict(ref64.named_parameters())
    floor = grad_floor(ref)
    ratios = []
    for name, p in m.named_parameters():
        t = p64[name].grad
        e_hip = (p.grad.detach().cpu().double() - t).abs().max().item()
        e_t32 = (ref_params[name].grad.double() - t).abs().max().item()
        ratios.append((e_hip / max(e_t32, 1e-3 * floor), name))
        assert e_hip <= 6 * e_t32 + 1e-2 * floor, f"{name}: |hip-f64|={e_hip:.3e} vs |torch32-f64|={e_t32:.3e}"
    print("worst (|hip-f64| / |torch32-f64|):", max(ratios))
    bufs = dict(m.named_buffers())
    np.testing.assert_allclose(bufs["feature_extractor.bn1.running_var"].cpu().numpy(), g["bn1_running_var"], rtol=1e-5)
    np.testing.assert_allclose(bufs["feature_extractor.layer4.1.bn2.running_var"].cpu().numpy(), g["l4_bn2_running_var"], rtol=1e-4)
    print("worst grad rel err", worst)


def test_training_step_ragged_shapes(dev, seeded_sd):
    """Edge shapes through the whole step: odd batch, non-square images whose maps shrink to odd sizes (96x80 ->
    48x40 -> 24x20 -> 12x10 -> 6x5 -> 3x3), i.e. ragged stem tiles, clipped pool windows, odd stride-2 parity classes,
    rows < one tile; and a 48x48 batch that goes through the nearest-resize branch.  Loss + every gradient vs autograd."""
    from self_supervised import training, ops
    from oracle.peranet import train_step
    for (b, h, w, seed) in [(5, 96, 80, 71), (8, 48, 48, 72)]:
        ref, m = _pair(seeded_sd, dev)
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(b, 3, h, w, generator=g)
        y = torch.randint(0, 4, (b,), generator=g)
        loss_ref, _, out_ref = train_step(ref, x, y)
        loss_ref.backward()
        m.unfreeze()
        eng = training.DataParallelStep(m, lr=0.03, world_size=1).eng
        logits, emb = eng.forward(x.to(dev))
        # BatchNorm1d over 3-5 rows divides by a tiny batch variance: fp32 summation order alone shows at ~1e-4
        assert rel_err(logits, out_ref["classifier"]) < 5e-4 and rel_err(emb, out_ref["latent_space"]) < 5e-4
        dlogits = torch.empty_like(logits)
        la = ops.softmax_ce(logits, y.to(dev), dlogits, 1.0 / b)
        np.testing.assert_allclose(la[0].item(), loss_ref.item(), rtol=1e-4)
        eng.backward(dlogits)
        ref_params = dict(ref.named_parameters())
        floor = grad_floor(ref)
        for name, p in m.named_parameters():
            # tiny batches make the BN chain ill-conditioned: fp32 summation order alone moves gradients by ~1e-3
            e = rel_err(p.grad, ref_params[name].grad, floor)
            assert e < 5e-3, f"{(b, h, w)} {name}: grad rel err {e:.3e}"


def test_two_sgd_steps_match_reference(dev, golden, seeded_sd):
    from self_supervised import training
    from oracle import weights as ow
    g = golden("train_step")
    _, m = _pair(seeded_sd, dev)
    m.unfreeze()
    step = training.DataParallelStep(m, lr=0.03, world_size=1)
    x, y = ow.synthetic_images(8, 64, seed=55).to(dev), ow.synthetic_labels(8, seed=56).to(dev)
    la = step.step(x, y)
    sd = m.state_dict()
    np.testing.assert_allclose(sd["classifier.weight"].cpu().numpy(), g["post_step_classifier_weight"], rtol=1e-4, atol=1e-6)
    # conv1 is the deepest gradient (fp32 noise of 20 layers of batch-8 BN backward): lr * 1e-3 * |g|max ~ 1e-5
    np.testing.assert_allclose(sd["feature_extractor.conv1.weight"][:4].cpu().numpy(), g["post_step_conv1_slice"], rtol=1e-4, atol=2e-5)
    la2 = step.step(x, y)
    np.testing.assert_allclose(la2[0].item(), float(g["loss2"]), rtol=1e-3)
    np.testing.assert_allclose(m.state_dict()["classifier.weight"].cpu().numpy(), g["post_step2_classifier_weight"], rtol=1e-3, atol=5e-5)


def test_frozen_backbone_stage(dev, seeded_sd):
    """Stage 1 of tools.training: backbone frozen, only the head gets gradients / updates."""
    from self_supervised import training
    from oracle import weights as ow
    from oracle.peranet import train_step
    ref, m = _pair(seeded_sd, dev)
    m.freeze_net(['backbone']); m.train()          # PL calls model.train() at fit start (quirk Q6)
    for p in ref.feature_extractor.parameters():
        p.requires_grad = False
    x, y = ow.synthetic_images(8, 64, seed=55), ow.synthetic_labels(8, seed=56)
    loss_ref, _, _ = train_step(ref, x, y)
    loss_ref.backward()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    step = training.DataParallelStep(m, lr=0.03, world_size=1)
    la = step.step(x.to(dev), y.to(dev))
    np.testing.assert_allclose(la[0].item(), loss_ref.item(), rtol=1e-5)
    ref_params = dict(ref.named_parameters())
    for name, p in m.named_parameters():
        if name.startswith("feature_extractor"):
            assert torch.equal(p.detach(), before[name]), name
        else:
            assert rel_err(p.grad, ref_params[name].grad, grad_floor(ref)) < 1e-3, name


def test_trainable_batchnorm_weights_under_eval_statistics(dev, seeded_sd):
    """BatchNorm layers of the trunk in eval mode (running statistics) while their affine parameters -- and everything else -- train:
    the reference's autograd handles it (models.py:174-196 only toggles requires_grad; a user may un-freeze any subset), the engine
    keeps z for such a layer and takes the weight gradient from the normalised input (round 4 raised NotImplementedError here)."""
    from self_supervised import training
    from oracle import weights as ow
    from oracle.peranet import train_step
    ref, m = _pair(seeded_sd, dev)
    m.unfreeze()
    for mod in list(ref.feature_extractor.modules()) + list(m.feature_extractor.modules()):
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eval()
    x, y = ow.synthetic_images(8, 64, seed=65), ow.synthetic_labels(8, seed=66)
    loss_ref, _, _ = train_step(ref, x, y)
    loss_ref.backward()
    step = training.DataParallelStep(m, lr=0.01, world_size=1, graph=False)
    la = step.step(x.to(dev), y.to(dev))
    np.testing.assert_allclose(la[0].item(), loss_ref.item(), rtol=1e-5)
    ref_params = dict(ref.named_parameters())
    for name, p in m.named_parameters():
        assert rel_err(p.grad, ref_params[name].grad, grad_floor(ref)) < 1e-3, name
    for (n1, b1), (n2, b2) in zip(ref.named_buffers(), m.named_buffers()):          # running statistics untouched
        if "feature_extractor" in n1 and "num_batches" not in n1:
            assert torch.allclose(b1, b2.cpu()), n1


def test_eval_statistics_flag_does_not_leak_into_the_next_step(dev, seeded_sd):
    """a forward under eval-mode statistics with trainable BatchNorm weights marks its layers (`eval_stats`); the
    mark belongs to THAT forward only.  A forward without a backward (validation with gradients enabled), then model.train() and an
    eager step: the stem, the raw-output convs of layer1 and every block's second conv must take the batch-statistics backward
    again -- gradients equal to the oracle's."""
    from self_supervised import training
    from oracle import weights as ow
    from oracle.peranet import train_step
    ref, m = _pair(seeded_sd, dev)
    m.unfreeze()
    bns = [mod for mod in m.feature_extractor.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
    for mod in bns:
        mod.eval()
    x, y = ow.synthetic_images(8, 64, seed=65), ow.synthetic_labels(8, seed=66)
    eng = training.get_engine(m)
    eng.forward(x.to(dev))                               # leaves eval_stats set on every trunk layer, tape not consumed
    assert eng.stem.eval_stats and eng.blocks[0]["c2"].eval_stats
    for mod in bns:
        mod.train()
    loss_ref, _, _ = train_step(ref, x, y)
    loss_ref.backward()
    step = training.DataParallelStep(m, lr=0.01, world_size=1, graph=False)
    la = step.step(x.to(dev), y.to(dev))
    assert not eng.stem.eval_stats and not any(d[k].eval_stats for d in eng.blocks for k in ("c1", "c2"))
    np.testing.assert_allclose(la[0].item(), loss_ref.item(), rtol=1e-5)
    ref_params = dict(ref.named_parameters())
    for name, p in m.named_parameters():
        assert rel_err(p.grad, ref_params[name].grad, grad_floor(ref)) < 1e-3, name


def test_side_branches_leave_the_step_bit_identical(dev, seeded_sd):
    """ops.ASIDE: slab reductions, head weight gradients, pooling rows and the backward pass's filter tables run as parallel branches of
    the step (a second stream; fork / join nodes of the recorded graph).  Same kernels, same operands: parameters, momentum and
    BatchNorm buffers after four steps (eager, capture, two replays) equal those of the single-chain step bit for bit, in fp32 and
    with half tensors; a replayed step really contains the branches."""
    from self_supervised import training
    from oracle import weights as ow
    x, y = ow.synthetic_images(8, 64, seed=81).to(dev), ow.synthetic_labels(8, seed=82).to(dev)
    for prec in (32, 16):
        states = []
        for aside in (False, True):
            _, m = _pair(seeded_sd, dev)
            m.unfreeze()
            step = training.DataParallelStep(m, lr=0.01, world_size=1, precision=prec)
            step.eng.sw_aside = aside
            for _ in range(4):
                step.step(x, y)
            torch.cuda.synchronize()
            assert step._plans, "the step was never recorded"
            assert (step.eng.aside.stream is not None) == aside
            if aside:
                assert ("flip32",) in step.eng._tables_used
            states.append(torch.cat([step.eng.arena.p, step.eng.arena.m] + [b.detach().flatten().float() for b in m.buffers()]).clone())
        assert torch.equal(states[0], states[1]), prec


def test_batched_slab_reductions_leave_the_step_bit_identical(dev, seeded_sd):
    """ops.PENDING_REDUCE: the slab reductions of a step's weight-gradient kernels run in ONE launch where the gradients are first needed
    (ssad_wgrad_reduce_batch) instead of one launch per layer.  Same order of additions per output: parameters, momentum and buffers
    after four steps equal the per-layer form bit for bit (fp32 and half tensors, eager and replayed)."""
    from self_supervised import ops, training
    from oracle import weights as ow
    x, y = ow.synthetic_images(8, 64, seed=83).to(dev), ow.synthetic_labels(8, seed=84).to(dev)
    for prec in (32, 16):
        states = []
        for batched in (False, True):
            _, m = _pair(seeded_sd, dev)
            m.unfreeze()
            step = training.DataParallelStep(m, lr=0.01, world_size=1, precision=prec)
            step.eng.sw_batch_reduce = batched
            for _ in range(4):
                step.step(x, y)
            torch.cuda.synchronize()
            assert step._plans and ops.PENDING_REDUCE is None
            states.append(torch.cat([step.eng.arena.p, step.eng.arena.m] + [b.detach().flatten().float() for b in m.buffers()]).clone())
        assert torch.equal(states[0], states[1]), prec
    # the kernel against the per-layer launch: three reductions of different shapes in one table
    g = torch.Generator().manual_seed(5)
    slabs = [torch.randn(s, co, k, generator=g).to(dev) for s, co, k in ((37, 64, 576), (8, 128, 1152), (1, 64, 64))]
    want = []
    for sl in slabs:
        out = torch.empty(sl.shape[1] * sl.shape[2], device=dev)
        ops._wgrad_reduce(sl, out, sl.shape[0], sl.shape[1], sl.shape[2], 1, 1, sl.shape[2], False, False)
        want.append(out)
    ops.PENDING_REDUCE = []
    try:
        got = []
        for sl in slabs:
            out = torch.empty(sl.shape[1] * sl.shape[2], device=dev)
            ops._wgrad_reduce(sl, out, sl.shape[0], sl.shape[1], sl.shape[2], 1, 1, sl.shape[2], False, False)
            got.append(out)
        assert len(ops.PENDING_REDUCE) == 3
        ops.flush_reductions()
    finally:
        ops.PENDING_REDUCE = None
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_fused_sgd_state_roundtrip_is_layout_independent(dev, seeded_sd):
    """FusedSGD.state_dict() holds the momentum WITHOUT the arena's alignment pads (round 6 pads every parameter to 32 bytes; the
    classifier's 4-element bias leaves a gap): what rounds 1-5 wrote still loads, a padded arena image loads too, and a reloaded
    optimizer continues bit-identically."""
    from self_supervised import training
    from oracle import weights as ow
    x, y = ow.synthetic_images(4, 64, seed=91).to(dev), ow.synthetic_labels(4, seed=92).to(dev)
    _, m = _pair(seeded_sd, dev)
    m.unfreeze()
    step = training.DataParallelStep(m, lr=0.01, world_size=1, graph=False)
    step.step(x, y); step.step(x, y)
    a = step.eng.arena
    assert a.total > sum(p.numel() for p in m.parameters()) and a.total % 8 == 0
    assert all(off % 8 == 0 for off, _ in a.offset.values())
    sd = step.opt.state_dict()
    assert sd["momentum"].numel() == sum(p.numel() for p in m.parameters())
    keep = a.m.clone()
    a.m.zero_()
    step.opt.load_state_dict(sd)
    assert torch.equal(a.m, keep)
    a.m.zero_()
    step.opt.load_state_dict({"param_groups": sd["param_groups"], "momentum": keep.cpu()})
    assert torch.equal(a.m, keep)


def test_bound_step_inputs(dev, seeded_sd):
    """DataParallelStep.bind_inputs: a producer fills the recorded step's own input buffers in place; replays from them equal replays
    that copy the batch in, bit for bit."""
    from self_supervised import training
    from oracle import weights as ow
    x, y = ow.synthetic_images(4, 64, seed=75).to(dev), ow.synthetic_labels(4, seed=76).to(dev)
    x2 = ow.synthetic_images(4, 64, seed=77).to(dev)
    outs = []
    for bound in (False, True):
        _, m = _pair(seeded_sd, dev)
        m.unfreeze()
        step = training.DataParallelStep(m, lr=0.01, world_size=1)
        assert step.bind_inputs(x, y)[0] is x                       # no plan yet: the tensors themselves
        for _ in range(3):
            step.step(x, y)
        if bound:
            xb, yb = step.bind_inputs(x, y)
            assert xb.data_ptr() != x.data_ptr() and torch.equal(xb, x)
            xb.copy_(x2)                                            # the producer writes the next batch in place
            step.step(xb, yb)
        else:
            step.step(x2, y)
        torch.cuda.synchronize()
        outs.append(step.eng.arena.p.clone())
    assert torch.equal(outs[0], outs[1])


def test_autograd_bridge(dev, seeded_sd):
    """loss.backward() on the tensor returned by training_step fills p.grad (PyTorch-Lightning-style use)."""
    from oracle import weights as ow
    from oracle.peranet import train_step
    ref, m = _pair(seeded_sd, dev)
    m.unfreeze()
    x, y = ow.synthetic_images(8, 64, seed=55), ow.synthetic_labels(8, seed=56)
    loss_ref, _, _ = train_step(ref, x, y)
    loss_ref.backward()
    loss = m.training_step((x.to(dev), y.to(dev), None), 0)
    loss.backward()
    ref_params = dict(ref.named_parameters())
    for name, p in m.named_parameters():
        assert rel_err(p.grad, ref_params[name].grad, grad_floor(ref)) < 1e-3, name
    (opt,), _ = m.configure_optimizers()
    opt.step()


def _bf(t):
    return t.bfloat16().float()


def test_bf16_operand_kernels(dev):
    """precision=16 path: conv fwd / dgrad / wgrad with operands rounded to bf16 == fp32 math on bf16-rounded inputs."""
    from self_supervised import ops
    for (n, h, w, cin, cout, k, s, p) in [(3, 8, 8, 64, 64, 3, 1, 1), (2, 9, 9, 64, 128, 3, 2, 1), (4, 6, 6, 128, 256, 3, 1, 1),
                                           (2, 8, 8, 64, 128, 1, 2, 0), (300, 1, 1, 896, 512, 1, 1, 0), (64, 16, 16, 160, 64, 1, 1, 0)]:
        g = torch.Generator().manual_seed(n + k)
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        xr, wr = _bf(x).requires_grad_(), _bf(wt).requires_grad_()
        y = F.conv2d(xr, wr, None, s, p)
        dy = torch.randn(y.shape, generator=g)
        nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
        w_ohwi = ops.repack_oihw_to_ohwi(wt.to(dev))
        got = ops.conv_fwd(nh(x), w_ohwi, None, None, None, False, s, p, bf16=True)
        assert rel_err(got.permute(0, 3, 1, 2), y) < 2e-5
        # backward operands are rounded too: reference = fp32 math on bf16(dy), bf16(w), bf16(x)
        dyr = _bf(dy)
        gx, = torch.autograd.grad(F.conv2d(xr, wr, None, s, p), xr, dyr)
        gw, = torch.autograd.grad(F.conv2d(xr, wr, None, s, p), wr, dyr)
        dx = ops.conv_dgrad(nh(dy), ops.flip_transpose_weight(w_ohwi), nh(x).shape, s, p, None, bf16=True)
        assert rel_err(dx.permute(0, 3, 1, 2), gx) < 2e-5
        dw = torch.empty(cout * k * k * cin, device=dev)
        ops.conv_wgrad(nh(dy), nh(x), dw, k, k, s, p, bf16=True)
        assert rel_err(dw.view(cout, k, k, cin).permute(0, 3, 1, 2), gw) < 2e-5


def test_bf16_training_matches_emulation(dev, seeded_sd):
    """Trainer(precision="bf16") (explicit opt-in; precision=16 is the fp16 path): loss and every gradient against the torch-CPU emulation (fp32 math on bf16-rounded
    operands in forward and backward, oracle/bf16_emul.py); and the loss goes down."""
    from self_supervised import training
    from oracle import weights as ow
    from oracle.peranet import train_step
    from oracle.bf16_emul import emulate_bf16
    ref, m = _pair(seeded_sd, dev)
    emulate_bf16(ref)
    x, y = ow.synthetic_images(16, 64, seed=55), ow.synthetic_labels(16, seed=56)
    loss_ref, _, _ = train_step(ref, x, y)
    loss_ref.backward()
    m.unfreeze()
    step = training.DataParallelStep(m, lr=0.01, world_size=1, precision="bf16")
    assert step.eng.bf16 is True
    la = step.step(x.to(dev), y.to(dev))
    np.testing.assert_allclose(la[0].item(), loss_ref.item(), rtol=1e-2)   # bf16 rounding boundaries flip with summation order
    ref_params = dict(ref.named_parameters())
    floor = grad_floor(ref)
    flat_h, flat_r = [], []
    for name, p in m.named_parameters():
        flat_h.append(p.grad.detach().cpu().flatten()); flat_r.append(ref_params[name].grad.flatten())
    cos = torch.nn.functional.cosine_similarity(torch.cat(flat_h), torch.cat(flat_r), dim=0).item()
    # Every kernel is exact against this emulation in isolation (test_bf16_operand_kernels).  End to end, 20 layers of
    # batch-16 BatchNorm on random weights make the bf16 chain chaotic at the 1e-2 level: an activation that sits on a
    # bf16 rounding boundary flips with the fp32 summation order.  Measured: cosine 0.96 against the emulation, 0.90
    # against the fp32 step -- the HIP path follows the emulated arithmetic, not just "something near fp32".
    ref32, _ = _pair(seeded_sd, dev)
    l32, _, _ = train_step(ref32, x, y)
    l32.backward()
    p32 = dict(ref32.named_parameters())
    flat_32 = torch.cat([p32[n].grad.flatten() for n, _ in m.named_parameters()])
    cos32 = torch.nn.functional.cosine_similarity(torch.cat(flat_h), flat_32, dim=0).item()
    assert cos > 0.93 and cos > cos32, (cos, cos32)
    first = la[0].item()
    for _ in range(8):
        last = step.step(x.to(dev), y.to(dev))[0].item()
    assert last < first


def test_layer1_in_layer_outputs(dev):
    """layer_outputs=['layer1','layer2','layer3'] (960-d concat, models.py:119-132): eval forward and training gradients."""
    from self_supervised.models import PeraNet
    from self_supervised import training
    from oracle import weights as ow
    from oracle.peranet import OraclePeraNet, train_step
    lo = ['layer1', 'layer2', 'layer3']
    sd = ow.seeded_state_dict(3, layer_outputs=lo)
    ref = OraclePeraNet(layer_outputs=lo); ref.load_state_dict(sd)
    m = PeraNet(layer_outputs=lo); m.load_state_dict(sd); m.to(dev)
    x, y = ow.synthetic_images(8, 64, seed=21), ow.synthetic_labels(8, seed=22)
    ref.eval(); m.eval()
    with torch.no_grad():
        want, got = ref(x), m(x.to(dev))
    assert rel_err(got["latent_space"], want["latent_space"]) < 1e-4 and rel_err(got["classifier"], want["classifier"]) < 1e-4
    ref.train(); m.train(); m.unfreeze()
    loss_ref, _, _ = train_step(ref, x, y)
    loss_ref.backward()
    step = training.DataParallelStep(m, lr=0.01, world_size=1)
    la = step.step(x.to(dev), y.to(dev))
    np.testing.assert_allclose(la[0].item(), loss_ref.item(), rtol=1e-5)
    ref_params, floor = dict(ref.named_parameters()), grad_floor(ref)
    for name, p in m.named_parameters():
        assert rel_err(p.grad, ref_params[name].grad, floor) < 1e-3, name


def test_deep_projection_head(dev):
    """PeraNet(latent_space_layers=16), a public constructor argument (models.py:26, :65-88): 15 linear layers in the latent MLP,
    so the step's batched weight flip holds 36 filters -- more than one launch of ssad_flip_transpose_batch takes."""
    from self_supervised.models import PeraNet
    from self_supervised import ops, training
    from oracle import weights as ow
    from oracle.peranet import OraclePeraNet, train_step
    sd = ow.seeded_state_dict(5, latent_space_layers=16)
    ref = OraclePeraNet(latent_space_layers=16); ref.load_state_dict(sd)
    m = PeraNet(latent_space_layers=16); m.load_state_dict(sd); m.to(dev)
    assert len(m.latent_space) == 16 and len(list(ref.latent_space)) == 16
    x, y = ow.synthetic_images(16, 64, seed=23), ow.synthetic_labels(16, seed=24)
    ref.train(); m.train(); m.unfreeze()
    loss_ref, _, _ = train_step(ref, x, y)
    loss_ref.backward()
    step = training.DataParallelStep(m, lr=0.01, world_size=1)
    la = step.step(x.to(dev), y.to(dev))
    assert step.eng._flip_tables[False]["n"] > 32
    # sixteen train-mode BatchNorm1d layers over 16 rows amplify summation-order differences: 1.3e-5 on the loss measured
    # (the five-layer head holds 1e-5), hence the wider bars of this one test
    np.testing.assert_allclose(la[0].item(), loss_ref.item(), rtol=1e-4)
    # gradients: a ReLU whose pre-activation sits within rounding of zero takes the other branch in one of the two implementations
    # (16 rows: one flipped unit moves its weight row by ~1/4, profiles/r02_relu_kink_evidence.md), so the bar is on the whole
    # gradient vector, and the flipped weights themselves are checked exactly below
    g_hip = torch.cat([p.grad.detach().flatten().cpu() for _, p in m.named_parameters()])
    ref_params = dict(ref.named_parameters())
    g_ref = torch.cat([ref_params[n].grad.flatten() for n, _ in m.named_parameters()])
    cos = torch.nn.functional.cosine_similarity(g_hip, g_ref, dim=0).item()
    assert cos > 0.98, cos
    # the batched flip (two launches: 32 + 4 filters) against the one-filter kernel, every filter, exactly
    eng = step.eng
    layers = [d[k] for d in eng.blocks for k in ("c1", "c2", "ds") if d[k] is not None] + list(eng.head) + [eng.cls]
    assert len(layers) == eng._flip_tables[False]["n"] == 36
    eng._flip_ready = False                  # flip the CURRENT weights (the step above has already updated them)
    eng.flipped(layers[0].lin, layers[0].weight())
    for layer in layers:
        w = layer.weight()
        assert torch.equal(eng._flip_tables[False]["view"][id(layer.lin.weight)], ops.flip_transpose_weight(w.contiguous())), layer.lin
    first = la[0].item()
    for _ in range(6):                       # graph capture + replay with the two-launch flip
        la = step.step(x.to(dev), y.to(dev))
    assert np.isfinite(la[0].item()) and la[0].item() < first


def test_rccl_bucketed_allreduce_single_rank(dev, seeded_sd):
    """The N > 1 code path on real RCCL: a one-rank "nccl" group, world_size forced to 2 so that the bucket hooks are
    live -- async all-reduces of arena prefixes interleaved with the backward kernels on the compute stream, waits,
    then the SGD kernel with the 1/2 scale.  A one-rank sum is the identity, so the gradients must equal the plain step's."""
    import os, socket
    import torch.distributed as dist
    from oracle import weights as ow
    from self_supervised import training
    from self_supervised.models import PeraNet
    x, y = ow.synthetic_images(8, 64, seed=55).to(dev), ow.synthetic_labels(8, seed=56).to(dev)
    m0 = PeraNet(); m0.load_state_dict(seeded_sd); m0.to(dev).train(); m0.unfreeze()
    s0 = training.DataParallelStep(m0, lr=0.03, world_size=1)
    s0.step(x, y)
    g0 = s0.eng.arena.g.clone()
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        m1 = PeraNet(); m1.load_state_dict(seeded_sd); m1.to(dev).train(); m1.unfreeze()
        s1 = training.DataParallelStep(m1, lr=0.03, world_size=2)
        la = s1.step(x, y)
        torch.cuda.synchronize()
        assert len(s1.bucketer.launched) >= 2                       # several buckets went out during backward
        covered = sum(e - b for b, e in s1.bucketer.launched)
        assert covered == s1.eng.arena.total
        assert torch.equal(s1.eng.arena.g, g0)                      # identity all-reduce, same deterministic kernels
        assert torch.isfinite(la[0]).item()
        # the update used grad / 2: p1 = p - lr * (g / 2 + wd * p)  vs  p0 = p - lr * (g + wd * p)
        p_init = torch.cat([seeded_sd[n].flatten() for n in ("classifier.weight",)]).to(dev)
        w0 = dict(m0.named_parameters())["classifier.weight"].detach().flatten()
        w1 = dict(m1.named_parameters())["classifier.weight"].detach().flatten()
        assert torch.allclose((p_init - w1) * 2 - 0.03 * 0.0005 * p_init, (p_init - w0), atol=1e-7)
        # hipGraph segments with REAL RCCL all-reduces issued between them (capture on step 2, replay on 3-4) == the same
        # steps launched eagerly
        m2 = PeraNet(); m2.load_state_dict(seeded_sd); m2.to(dev).train(); m2.unfreeze()
        s2 = training.DataParallelStep(m2, lr=0.03, world_size=2, graph=False)
        s2.step(x, y)
        for _ in range(3):
            s1.step(x, y); s2.step(x, y)
        torch.cuda.synchronize()
        plan = next(iter(s1._plans.values()))
        assert sum(1 for o in plan["ops"] if o[0] == "allreduce") >= 2 and sum(1 for o in plan["ops"] if o[0] == "graph") >= 3
        # the replays were watched: an event behind every plan op, every step seen to finish (StepWatchdog)
        assert s1.watchdog is not None and len(plan["events"][0]) == len(plan["ops"]) and plan["ev_step"] >= 2
        assert s1.watchdog.check() is None and all(e.query() for e in plan["events"][(plan["ev_step"] - 1) % 4])
        assert torch.equal(s1.eng.arena.p, s2.eng.arena.p) and torch.equal(s1.eng.arena.m, s2.eng.arena.m)
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
# round 2: kernels at the grid sizes bench.py times, the step at the benchmark batch, fp16 operands
# ---------------------------------------------------------------------------------------------
def test_large_grid_conv_dgrad_wgrad(dev):
    """Shapes whose 128x128 tiling has >= 500 workgroups, i.e. the `launch<128,128,2,2,32>` fp32 instantiation that the
    bs256 benchmark runs on layers 2-4 (the small-grid switch sends every smaller case to the 128x64 tile): plain conv,
    the BN-statistics epilogue, dgrad (stride 1 and the parity-class stride-2 form) and the cost-model wgrad splits."""
    from self_supervised import ops
    for (n, h, cin, cout, k, s, p) in [(64, 32, 128, 128, 3, 1, 1), (128, 32, 128, 256, 3, 2, 1), (160, 16, 256, 256, 3, 1, 1)]:
        g = torch.Generator().manual_seed(n + cout)
        x = torch.randn(n, cin, h, h, generator=g, requires_grad=True)
        wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).requires_grad_()
        y = F.conv2d(x, wt, None, s, p)
        ho = y.shape[-1]
        assert ((n * ho * ho + 127) // 128) * ((cout + 127) // 128) >= 500
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
        w_ohwi = ops.repack_oihw_to_ohwi(wt.detach().to(dev))
        got = ops.conv_fwd(nh(x), w_ohwi, None, None, None, False, s, p)
        assert rel_err(got.permute(0, 3, 1, 2), y) < 2e-5
        rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
        z2, mean, invstd = ops.conv_fwd_stats(nh(x), w_ohwi, 1e-5, 0.1, rm, rv, s, p)
        assert torch.equal(z2, got)
        yd = y.detach().double()
        assert rel_err(mean, yd.mean((0, 2, 3))) < 1e-5
        assert rel_err(invstd, (yd.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt()) < 1e-5
        assert rel_err(rv, 0.9 + 0.1 * yd.var((0, 2, 3), unbiased=True)) < 1e-5
        dx = ops.conv_dgrad(nh(dy), ops.flip_transpose_weight(w_ohwi), nh(x).shape, s, p)
        assert rel_err(dx.permute(0, 3, 1, 2), x.grad) < 2e-5
        dw = torch.empty(cout * k * k * cin, device=dev)
        ops.conv_wgrad(nh(dy), nh(x), dw, k, k, s, p)
        assert rel_err(dw.view(cout, k, k, cin).permute(0, 3, 1, 2), wt.grad) < 2e-5
        dw2 = torch.empty_like(dw)
        ops.conv_wgrad(nh(dy), nh(x), dw2, k, k, s, p)
        assert torch.equal(dw, dw2)                                   # fixed-order slab reduction


def _flat_grads(module, names):
    p = dict(module.named_parameters())
    return [p[n].grad.detach().cpu().double().flatten() for n in names]


class _FixedMaskReLU(torch.nn.Module):
    """ReLU with its active set prescribed: y = z * mask, dz = dy * mask (the branch the HIP forward took)."""

    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, z):
        return z * self.mask.to(z.dtype)


@pytest.mark.parametrize("batch", [32, 256])
def test_training_step_at_benchmark_size(dev, seeded_sd, batch):
    """BASELINE configs[1] (and the per-rank batch of configs[2]): one training step of (batch, 3, 256, 256).

    Forward (logits, embeddings 1e-4), loss (1e-5) and BatchNorm running statistics (1e-4) are held against torch-CPU
    fp32 autograd on the oracle exactly as in the small-shape test.  Gradients cannot be held per element at this size
    by ANY fp32 implementation: a ReLU whose pre-activation lies within fp32 noise of zero (|bn(z)| ~ 1e-6: tens of them
    among the 1e8 activations of a step) takes the other branch.  In the trunk one flipped element is one of >= 2048
    positions of its channel; in the projection head it is one of `batch` rows, moves a whole row of a weight gradient by
    ~1/sqrt(batch) and changes the signal sent down to every layer below (measured: feeding the ORACLE head the HIP
    trunk's pooled features, 4.5e-6 away from its own, flips one unit and reproduces the 1.66e-1 deviation of
    `latent_space.0.0.weight` digit for digit -- tools/diag_head3.py, profiles/r02_relu_kink_evidence.md).  Therefore
    (i) the three head ReLUs of the oracle are given the active set the HIP forward chose (each disagreement must be a
    genuine kink: oracle pre-activation within 1e-4 of zero), and (ii) the yardstick is an fp64 run: the HIP gradient must
    be as close to it as torch-CPU fp32 is -- globally within 2x in relative L2, per tensor within 4x (+1e-4) for EVERY
    tensor (measured: 78 / 78), and NO tensor beyond 5e-2 (a wrong tap / mask / reduction is O(1)).
    Plus bit-identical repeat runs, eagerly and as a replayed hipGraph."""
    import copy
    from self_supervised import training, ops
    from oracle import weights as ow
    from oracle.peranet import train_step
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    ref, m = _pair(seeded_sd, dev)
    x, y = ow.synthetic_images(batch, 256, seed=1234), ow.synthetic_labels(batch, seed=1235)
    m.unfreeze()
    step = training.DataParallelStep(m, lr=0.005, world_size=1, graph=False)
    eng = step.eng
    xd, yd = x.to(dev), y.to(dev)
    logits, emb = eng.forward(xd)
    masks = []
    for layer in eng.head:
        if layer.relu:                                      # the branch each head unit took in the HIP forward
            bn = layer.bn
            pre = (layer.z.view(batch, -1) - layer.mean) * layer.invstd * bn.weight.detach() + bn.bias.detach()
            masks.append((pre > 0).float().cpu())
    assert len(masks) == 3
    dlogits = torch.empty_like(logits)
    la = ops.softmax_ce(logits, yd, dlogits, 1.0 / batch)
    eng.backward(dlogits)

    # oracle, first as it is (forward parity + how many head units sit on a kink), then with the HIP active set
    pre_ref = []
    hooks = [ref.latent_space[i][1].register_forward_hook(lambda mod, inp, out: pre_ref.append(out.detach())) for i in range(3)]
    with torch.no_grad():
        out_plain = ref(x)
    for h in hooks:
        h.remove()
    flips = 0
    for pr, mk in zip(pre_ref, masks):
        dis = (pr > 0).float() != mk
        flips += int(dis.sum())
        assert (pr[dis].abs() < 1e-4).all(), "a head unit far from its kink took a different branch"
    print(f"batch {batch}: {flips} head units on a ReLU kink took the other branch in the HIP forward")
    assert flips <= 16
    assert rel_err(logits, out_plain["classifier"]) < 1e-4 and rel_err(emb, out_plain["latent_space"]) < 1e-4
    ref.load_state_dict(seeded_sd)                          # undo the running-statistics update of the probe forward
    for i in range(3):
        ref.latent_space[i][2] = _FixedMaskReLU(masks[i])
    ref64 = copy.deepcopy(ref).double()
    loss_ref, _, _ = train_step(ref, x, y)            # BN statistics need the whole batch: no chunking (~30 GB host RAM at 256)
    loss_ref.backward()
    names = [n for n, _ in ref.named_parameters()]
    g32 = _flat_grads(ref, names)
    np.testing.assert_allclose(la[0].item(), loss_ref.item(), rtol=1e-5)
    want_buf = {k: v.clone() for k, v in ref.named_buffers()}
    del loss_ref
    ref.zero_grad(set_to_none=True)
    l64, _, _ = train_step(ref64, x.double(), y)
    l64.backward()
    g64 = _flat_grads(ref64, names)
    del ref64, l64

    mb = dict(m.named_buffers())
    for name in ("feature_extractor.bn1.running_var", "feature_extractor.layer1.0.bn1.running_mean",
                 "feature_extractor.layer4.1.bn2.running_var", "latent_space.4.running_var"):
        assert rel_err(mb[name], want_buf[name]) < 1e-4, name
    gh = _flat_grads(m, names)
    cat = torch.cat
    n64 = cat(g64).norm().item()
    E_hip, E_t32 = (cat(gh) - cat(g64)).norm().item() / n64, (cat(g32) - cat(g64)).norm().item() / n64
    print(f"batch {batch}: relative L2 distance to the fp64 gradient: HIP {E_hip:.3e}, torch-CPU fp32 {E_t32:.3e}")
    assert E_hip <= 2 * E_t32 + 1e-6, (E_hip, E_t32)
    floor = 1e-4 * max(t.abs().max().item() for t in g64)
    within, worst = 0, (0.0, "")
    for n, a, b, c in zip(names, gh, g32, g64):
        den = max(c.norm().item(), floor * c.numel() ** 0.5)          # analytically-zero gradients are rounding noise
        e_h, e_t = (a - c).norm().item() / den, (b - c).norm().item() / den
        worst = max(worst, (e_h, n))
        assert e_h < 5e-2, f"batch {batch} {n}: relative L2 error {e_h:.3e} (torch-CPU fp32: {e_t:.3e})"
        within += e_h <= 4 * e_t + 1e-4
    print(f"batch {batch}: {within}/{len(names)} tensors within 4x of torch-CPU fp32's own distance to fp64; worst {worst}")
    assert within == len(names), (within, len(names), worst)
    g_first = eng.arena.g.clone()
    # determinism: the same step again from the same state, eagerly and as a replayed hipGraph
    _, m2 = _pair(seeded_sd, dev)
    m2.unfreeze()
    s2 = training.DataParallelStep(m2, lr=0.005, world_size=1, graph=False)
    s2.step(xd, yd)
    assert torch.equal(s2.eng.arena.g, g_first)
    _, m3 = _pair(seeded_sd, dev)
    m3.unfreeze()
    s3 = training.DataParallelStep(m3, lr=0.005, world_size=1, graph=True)
    for _ in range(3):
        s3.step(xd, yd)                                                # eager, capture + replay, replay
    s2.step(xd, yd); s2.step(xd, yd)
    assert len(s3._plans) == 1
    assert torch.equal(s3.eng.arena.p, s2.eng.arena.p) and torch.equal(s3.eng.arena.m, s2.eng.arena.m)
    for (n2, b2), (n3, b3) in zip(m2.named_buffers(), m3.named_buffers()):
        assert torch.equal(b2, b3), n2


def _h(t):
    return t.half().float()


def test_f16_operand_kernels(dev):
    """precision=16 path: conv fwd / dgrad / wgrad with operands rounded to fp16 == fp32 math on fp16-rounded inputs
    (incl. a grid >= 500 case on the 128x128 tile)."""
    from self_supervised import ops
    for (n, h, w, cin, cout, k, s, p) in [(3, 8, 8, 64, 64, 3, 1, 1), (2, 9, 9, 64, 128, 3, 2, 1), (4, 6, 6, 128, 256, 3, 1, 1),
                                           (2, 8, 8, 64, 128, 1, 2, 0), (300, 1, 1, 896, 512, 1, 1, 0), (64, 16, 16, 160, 64, 1, 1, 0),
                                           (64, 32, 32, 128, 128, 3, 1, 1)]:
        g = torch.Generator().manual_seed(n + k)
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        xr, wr = _h(x).requires_grad_(), _h(wt).requires_grad_()
        y = F.conv2d(xr, wr, None, s, p)
        dy = torch.randn(y.shape, generator=g)
        nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
        w_ohwi = ops.repack_oihw_to_ohwi(wt.to(dev))
        got = ops.conv_fwd(nh(x), w_ohwi, None, None, None, False, s, p, bf16=2)
        assert rel_err(got.permute(0, 3, 1, 2), y) < 2e-5
        dyr = _h(dy)
        gx, = torch.autograd.grad(F.conv2d(xr, wr, None, s, p), xr, dyr)
        gw, = torch.autograd.grad(F.conv2d(xr, wr, None, s, p), wr, dyr)
        dx = ops.conv_dgrad(nh(dy), ops.flip_transpose_weight(w_ohwi), nh(x).shape, s, p, None, bf16=2)
        assert rel_err(dx.permute(0, 3, 1, 2), gx) < 2e-5
        dw = torch.empty(cout * k * k * cin, device=dev)
        ops.conv_wgrad(nh(dy), nh(x), dw, k, k, s, p, bf16=2)
        assert rel_err(dw.view(cout, k, k, cin).permute(0, 3, 1, 2), gw) < 2e-5


def test_halo_conv_c64_with_16_bit_operands(dev):
    """csrc/conv_c64.hip, OP = 1 / 2 (layer1 convolutions of the precision-16 step): fp32 math on bf16- / fp16-rounded operands; the input
    transform (BatchNorm + ReLU of the producer) and the emitted activation stay fp32 and are rounded after; residual, statistics and
    the input-gradient use (flipped filter) as in the fp32 kernel; ragged tiles."""
    from self_supervised import ops
    for (n, h, w) in [(3, 16, 16), (2, 13, 21), (5, 64, 64)]:
        g = torch.Generator().manual_seed(n * 10 + h)
        x = torch.randn(n, 64, h, w, generator=g)
        wt = torch.randn(64, 64, 3, 3, generator=g) / 24.0
        res = torch.randn(n, 64, h, w, generator=g)
        mean, var = torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5
        gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
        invstd = (var + 1e-5).rsqrt()
        nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
        w_ohwi = ops.repack_oihw_to_ohwi(wt.to(dev))
        xd, rd = nh(x), nh(res)
        for mode, rnd in ((2, _h), (1, lambda t: t.to(torch.bfloat16).float())):
            want = F.conv2d(rnd(x), rnd(wt), None, 1, 1)
            got = ops.conv3x3_c64(xd, w_ohwi, bf16=mode)
            assert rel_err(got.permute(0, 3, 1, 2), want) < 2e-5, (n, h, w, mode)
            got = ops.conv3x3_c64(xd, w_ohwi, residual=rd, bf16=mode)
            assert rel_err(got.permute(0, 3, 1, 2), want + res) < 2e-5
            act = torch.relu((x - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1))
            rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
            z, em, m2, i2 = ops.conv3x3_c64(xd, w_ohwi, transform=tuple(t.to(dev) for t in (mean, invstd, gamma, beta)), emit=True,
                                            stats=(1e-5, 0.1, rm, rv), bf16=mode)
            assert rel_err(em.permute(0, 3, 1, 2), act) < 1e-6                      # emitted in fp32, before the rounding
            want_t = F.conv2d(rnd(em.permute(0, 3, 1, 2).cpu()), rnd(wt), None, 1, 1)   # (rounded from the kernel's own fp32 activation)
            assert rel_err(z.permute(0, 3, 1, 2), want_t) < 2e-5
            zc = z.detach().cpu().double().reshape(-1, 64)
            assert (m2.cpu().double() - zc.mean(0)).abs().max() < 1e-5 and rel_err(i2, (zc.var(0, unbiased=False) + 1e-5).rsqrt()) < 1e-5
            # input gradient of the same layer: the flipped filter through the same kernel
            xr = rnd(x).requires_grad_()
            gx, = torch.autograd.grad(F.conv2d(xr, rnd(wt), None, 1, 1), xr, rnd(res))
            dx = ops.conv3x3_c64(rd, ops.flip_transpose_weight(w_ohwi), bf16=mode)
            assert rel_err(dx.permute(0, 3, 1, 2), gx) < 2e-5


def test_stem_conv_with_16_bit_operands(dev):
    """csrc/stem16.hip (conv1 of the precision-16 step): fp32 math on fp16- (bf16-) rounded image and weights, sizes that are ragged
    against the 8 x 32 output tile, BatchNorm statistics and running statistics from the raw output as the fp32 kernel takes them."""
    from self_supervised import ops
    for (b, h, w) in [(2, 64, 64), (3, 70, 74), (1, 256, 256), (5, 65, 97)]:
        g = torch.Generator().manual_seed(b * 100 + h)
        img = torch.randn(b, 3, h, w, generator=g) * 1.2
        wt = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
        for mode, rnd in ((2, _h), (1, lambda t: t.to(torch.bfloat16).float())):
            want = F.conv2d(rnd(img), rnd(wt), None, 2, 3)
            rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
            z, mean, invstd = ops.stem_fwd_stats16(img.to(dev), wt.to(dev), 1e-5, 0.1, rm, rv, mode)
            assert rel_err(z.permute(0, 3, 1, 2), want) < 2e-5, (b, h, w, mode)
            zc = z.detach().cpu().double().reshape(-1, 64)
            m, v = zc.mean(0), zc.var(0, unbiased=False)
            assert (mean.cpu().double() - m).abs().max() < 1e-5 and rel_err(invstd, (v + 1e-5).rsqrt()) < 1e-5
            n = zc.shape[0]
            assert rel_err(rm, 0.1 * m) < 1e-4 and rel_err(rv, 0.9 + 0.1 * v * n / max(n - 1, 1)) < 1e-4


def test_halo_wgrad_with_16_bit_operands(dev):
    """csrc/wgrad_halo16.hip (3x3 / stride 1 / pad 1 weight gradients of the precision-16 step): fp32 math on fp16- (bf16-) rounded
    operands, on maps that are ragged against both tile shapes (4 x 16 for widths above 8, 8 x 8 below), one and several (co, ci)
    blocks, one and many pixel tiles per workgroup; and the same numbers as the split-over-pixels kernel it replaces, to fp32
    summation order."""
    from self_supervised import _hip, ops
    lib = _hip.lib()
    for (n, h, w, cin, cout) in [(3, 8, 8, 64, 64), (2, 5, 7, 64, 128), (4, 6, 6, 128, 256), (2, 20, 20, 64, 64), (5, 16, 16, 128, 64),
                                 (64, 32, 32, 128, 128), (3, 13, 33, 64, 64), (40, 64, 64, 64, 64)]:
        assert lib.ssad_wgrad3x3_halo16_ok(cin, cout, 3, 3, 1, 1) == 1
        g = torch.Generator().manual_seed(n * 1000 + h)
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        dy = torch.randn(n, cout, h, w, generator=g)
        nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
        for mode, rnd in ((2, _h), (1, lambda t: t.to(torch.bfloat16).float())):
            xr, wr = rnd(x).requires_grad_(), rnd(wt).requires_grad_()
            gw, = torch.autograd.grad(F.conv2d(xr, wr, None, 1, 1), wr, rnd(dy))
            dw = torch.empty(cout * 9 * cin, device=dev)
            dyd, xd = nh(dy), nh(x)
            ops.conv_wgrad(dyd, xd, dw, 3, 3, 1, 1, bf16=mode)
            assert rel_err(dw.view(cout, 3, 3, cin).permute(0, 3, 1, 2), gw) < 2e-5, (n, h, w, cin, cout, mode)
            # the kernel it replaces (one tap per workgroup), called through the C ABI directly
            m = n * h * w
            splits = lib.ssad_wgrad_splits_bf16(m, cin, cout, 3, 3)
            slab = torch.empty((splits, cout, 9 * cin), device=dev)
            fn = lib.ssad_conv_wgrad_f16 if mode == 2 else lib.ssad_conv_wgrad_bf16
            _hip.check(fn(_hip.ptr(dyd), _hip.ptr(xd), _hip.ptr(slab), splits, n, h, w, cin, cout, 3, 3, 1, 1, dyd.numel(), _hip.stream()))
            old = torch.empty_like(dw)
            _hip.check(lib.ssad_wgrad_reduce(_hip.ptr(slab), _hip.ptr(old), splits, cout, 9 * cin, 3, 3, cin, 0, 0, _hip.stream()))
            assert rel_err(dw, old) < 2e-6, (n, h, w, cin, cout, mode)
    assert lib.ssad_wgrad3x3_halo16_ok(64, 128, 3, 3, 2, 1) == 0 and lib.ssad_wgrad3x3_halo16_ok(64, 64, 1, 1, 1, 0) == 0
    assert lib.ssad_wgrad3x3_halo16_ok(96, 64, 3, 3, 1, 1) == 0


def test_loss_scaler_state_machine(dev):
    """GradScaler semantics on the device: scale, inf detection, skipped update + backoff, growth after `interval` clean steps."""
    from self_supervised import ops
    from self_supervised.training import LossScaler
    sc = LossScaler(dev, init_scale=1024.0, growth_interval=2)
    d = torch.ones(8, device=dev)
    ops.scale_by_loss_scale(d, sc.state)
    assert torch.equal(d.cpu(), torch.full((8,), 1024.0))
    p, m = torch.ones(1000, device=dev), torch.zeros(1000, device=dev)
    g = torch.full((1000,), 1024.0, device=dev)
    hyper = torch.tensor([0.5, 0.9, 0.0, 1.0], device=dev)
    ops.check_finite(g, sc.state)
    assert sc.state[2].item() == 0.0
    ops.sgd_step_dev(p, g, m, hyper, sc.state)                      # unscaled gradient 1 -> p = 1 - 0.5
    assert torch.allclose(p.cpu(), torch.full((1000,), 0.5)) and torch.allclose(m.cpu(), torch.ones(1000))
    ops.loss_scaler_update(sc.state, 2.0, 0.5, 2)
    assert sc.state.tolist() == [1024.0, 1.0, 0.0]
    g[777] = float("inf")
    ops.check_finite(g, sc.state)
    assert sc.state[2].item() == 1.0
    before = p.clone()
    ops.sgd_step_dev(p, g, m, hyper, sc.state)                      # skipped
    assert torch.equal(p, before)
    ops.loss_scaler_update(sc.state, 2.0, 0.5, 2)
    assert sc.state.tolist() == [512.0, 0.0, 0.0]
    g[777] = float("nan"); ops.check_finite(g, sc.state); assert sc.state[2].item() == 1.0
    ops.loss_scaler_update(sc.state, 2.0, 0.5, 2)
    g[777] = 1.0
    for _ in range(2):
        ops.check_finite(g, sc.state); ops.loss_scaler_update(sc.state, 2.0, 0.5, 2)
    assert sc.state.tolist() == [512.0, 0.0, 0.0]                   # 256 -> grew back to 512 after 2 clean steps
    # hyper-parameters are read from device memory (a captured step must not bake them in)
    p2, m2 = torch.ones(10, device=dev), torch.zeros(10, device=dev)
    ops.sgd_step_dev(p2, torch.ones(10, device=dev), m2, torch.tensor([0.1, 0.9, 0.5, 2.0], device=dev), None)
    assert torch.allclose(p2.cpu(), torch.full((10,), 1 - 0.1 * (2.0 + 0.5)))


def test_precision16_step_on_images_below_and_above_64(dev, seeded_sd):
    """The 16-bit kernels of the precision-16 step choose themselves by shape: 32 x 32 images are resized to 64 x 64 by the fp32 stem's
    loader (models.py:217-219; the 16-bit stem takes whole images from 64 x 64 up), 96 x 72 ones take the 16-bit stem and ragged halo
    tiles.  Eager step, captured step, replayed step: the same finite loss going down, and the gradient of the 16-bit step close to the
    exact one (a 16-bit stem that mis-sized its output made the fp32 stem weight gradient read past its input: round 4)."""
    from self_supervised import training
    from oracle import weights as ow
    for size in ((32, 32), (96, 72)):
        g = torch.Generator().manual_seed(size[0])
        x = torch.randn(8, 3, *size, generator=g).to(dev)
        y = ow.synthetic_labels(8, seed=3).to(dev)
        grads = {}
        for prec in (32, 16):
            _, m = _pair(seeded_sd, dev)
            m.unfreeze()
            step = training.DataParallelStep(m, lr=0.01, world_size=1, precision=prec)
            losses = [float(step.step(x, y)[0])]
            grads[prec] = step.eng.arena.g.detach().clone()            # the first step's gradient: same weights in both precisions
            losses += [float(step.step(x, y)[0]) for _ in range(3)]
            torch.cuda.synchronize()
            assert all(np.isfinite(l) for l in losses) and losses[-1] < losses[0], (size, prec, losses)
        g16 = grads[16] / max(float(grads[16].abs().max()), 1e-30)      # (the 16-bit step's gradient carries the loss scale)
        cos = torch.nn.functional.cosine_similarity(g16, grads[32], dim=0).item()
        assert cos > 0.7, (size, cos)         # fp16 operands through 20 layers of batch-8 BatchNorm: direction, not digits


def test_f16_training_step_vs_autocast_oracle(dev, seeded_sd):
    """Trainer(precision=16) = fp16 operands + loss scaling, against the reference's own arithmetic: the oracle under
    torch.autocast(dtype=float16) with a scaled loss (what pl.Trainer(precision=16) runs, tools.py:263).  Autocast also
    rounds every activation to fp16 in HBM, the HIP path keeps fp32 tensors and rounds operands only, so the two are not
    bit-comparable; the a-priori bar is that the HIP gradient is at least as close to the exact fp32 gradient as the
    reference's arithmetic is (x1.25 slack for summation order), the loss within fp16 resolution, and the update finite."""
    from self_supervised import training
    from oracle import weights as ow
    from oracle.peranet import train_step
    x, y = ow.synthetic_images(16, 64, seed=55), ow.synthetic_labels(16, seed=56)
    ref32, m = _pair(seeded_sd, dev)
    l32, _, _ = train_step(ref32, x, y)
    l32.backward()
    g32 = torch.cat([p.grad.flatten() for p in ref32.parameters()])
    S = 65536.0
    while True:                                        # GradScaler: an overflowing step is skipped and the scale halved
        ref16, _ = _pair(seeded_sd, dev)
        with torch.autocast("cpu", dtype=torch.float16):
            l16, _, _ = train_step(ref16, x, y)
        (l16.float() * S).backward()
        g16 = torch.cat([p.grad.flatten() for p in ref16.parameters()]) / S
        if torch.isfinite(g16).all():
            break
        S /= 2
        assert S >= 1.0
    print(f"autocast oracle: loss scale {S:g} is the largest without fp16 overflow")
    m.unfreeze()
    step = training.DataParallelStep(m, lr=0.01, world_size=1, precision=16, graph=False)
    assert step.eng.bf16 == 2 and step.scaler is not None and step.scaler.get_scale() == 65536.0
    la = step.step(x.to(dev), y.to(dev))
    names = [n for n, _ in ref32.named_parameters()]
    hp = dict(m.named_parameters())
    gh = torch.cat([hp[n].grad.detach().cpu().flatten() for n in names]) / 65536.0      # the arena holds scaled gradients
    assert torch.isfinite(gh).all()
    e_hip = ((gh - g32).norm() / g32.norm()).item()
    e_ref = ((g16 - g32).norm() / g32.norm()).item()
    print(f"relative L2 distance to the fp32 gradient: HIP fp16 operands {e_hip:.4f}, autocast oracle {e_ref:.4f}")
    assert e_hip <= 1.25 * e_ref, (e_hip, e_ref)
    assert abs(la[0].item() - l32.item()) <= max(2 * abs(l16.item() - l32.item()), 2e-3 * abs(l32.item()))
    assert step.scaler.state.tolist() == [65536.0, 1.0, 0.0]        # finite step: tracker advanced, nothing skipped
    first = la[0].item()
    for _ in range(8):
        last = step.step(x.to(dev), y.to(dev))[0].item()
    assert last < first


def test_f16_training_step_vs_autocast_oracle_on_256px_images(dev, seeded_sd):
    """The same a-priori bar on the benchmark's image size (16 x 3 x 256 x 256: 64 x 64 .. 8 x 8 maps, i.e. the 8 x 16-tile and the
    two-maps-per-tile forms of csrc/conv16.hip, the halo-tile weight gradients on full maps, persistent workgroups walking several
    tiles), with the trunk's tensors stored as halves: the HIP gradient must be at least as close to the exact fp32 gradient as the
    reference's own arithmetic -- the oracle under torch.autocast(float16), which rounds every activation too -- is (x 1.25).
    (Batch 256 is not testable this way: fp16 autocast on the CPU takes ~4 s per image.)  A second step object with SSAD_ACT16's switch
    off (fp32 tensors, operands rounded while staged: rounds 2-4) must meet the same bar."""
    from self_supervised import training
    from oracle import weights as ow
    from oracle.peranet import train_step
    x, y = ow.synthetic_images(16, 256, seed=155), ow.synthetic_labels(16, seed=156)
    ref32, m = _pair(seeded_sd, dev)
    l32, _, _ = train_step(ref32, x, y)
    l32.backward()
    g32 = torch.cat([p.grad.flatten() for p in ref32.parameters()])
    S = 65536.0
    while True:
        ref16, _ = _pair(seeded_sd, dev)
        with torch.autocast("cpu", dtype=torch.float16):
            l16, _, _ = train_step(ref16, x, y)
        (l16.float() * S).backward()
        g16 = torch.cat([p.grad.flatten() for p in ref16.parameters()]) / S
        if torch.isfinite(g16).all():
            break
        S /= 2
        assert S >= 1.0
    e_ref = ((g16 - g32).norm() / g32.norm()).item()
    names = [n for n, _ in ref32.named_parameters()]
    for half in (True, False):
        _, mm = _pair(seeded_sd, dev)
        mm.unfreeze()
        step = training.DataParallelStep(mm, lr=0.01, world_size=1, precision=16, graph=False)
        step.eng.sw_act16 = half
        la = step.step(x.to(dev), y.to(dev))
        assert step.eng.h16 == half
        hp = dict(mm.named_parameters())
        gh = torch.cat([hp[n].grad.detach().cpu().flatten() for n in names]) / 65536.0
        assert torch.isfinite(gh).all()
        e_hip = ((gh - g32).norm() / g32.norm()).item()
        print(f"256 px, half tensors {half}: relative L2 distance to the fp32 gradient: HIP {e_hip:.4f}, autocast oracle {e_ref:.4f}")
        assert e_hip <= 1.25 * e_ref, (half, e_hip, e_ref)
        assert abs(la[0].item() - l32.item()) <= max(2 * abs(l16.item() - l32.item()), 2e-3 * abs(l32.item()))


def test_f16_training_step_at_batch_256(dev, seeded_sd):
    """the precision-16 step as a WHOLE at the benchmark's batch (256 x 3 x 256 x 256), where the register-fed conv, the
    transposing weight gradient and the nibble masks pick the forms the bench times.  The autocast oracle cannot run there (fp16 on
    the CPU: ~4 s per image), so the yardstick is the exact-fp32 step on the same batch -- itself held to torch fp64 at this size by
    test_training_step_at_benchmark_size: loss and every BatchNorm running statistic within fp16 resolution of the fp32 step's, the
    gradient no further from it than the 256-pixel oracle test allows at batch 16 (relative L2 0.25: measured there 0.20, half
    tensors and autocast alike), no non-finite value, and recorded replays bit-identical to each other."""
    from self_supervised import training
    from self_supervised.models import PeraNet
    from oracle import weights as ow
    x, y = ow.synthetic_images(256, 256, seed=255).to(dev), ow.synthetic_labels(256, seed=256).to(dev)

    def fresh():
        m = PeraNet(); m.load_state_dict(seeded_sd); m.to(dev).train(); m.unfreeze()
        return m
    m32 = fresh()
    s32 = training.DataParallelStep(m32, lr=0.01, world_size=1, precision=32, graph=False)
    l32 = s32.step(x, y)[0].item()
    g32 = s32.eng.arena.g.clone()
    m16 = fresh()
    s16 = training.DataParallelStep(m16, lr=0.01, world_size=1, precision=16, graph=False)
    l16 = s16.step(x, y)[0].item()
    assert s16.eng.h16
    g16 = s16.eng.arena.g / 65536.0
    assert torch.isfinite(g16).all()
    e = ((g16 - g32).norm() / g32.norm()).item()
    print(f"batch 256: precision-16 gradient against the fp32 step's: relative L2 {e:.4f}; loss {l16:.6f} vs {l32:.6f}")
    assert e <= 0.25, e
    assert abs(l16 - l32) <= 2e-3 * abs(l32)
    b32, b16 = dict(m32.named_buffers()), dict(m16.named_buffers())
    for name, t in b32.items():
        if "num_batches" in name:
            assert torch.equal(t, b16[name])
            continue
        d = (b16[name] - t).abs().max().item()
        assert d <= 2e-3 * max(1.0, t.abs().max().item()), (name, d)
    del s32, m32, g32
    torch.cuda.empty_cache()
    # recorded replays: two runs of (eager, capture, two replays) from the same state end in the same bits
    ends = []
    for _ in range(2):
        m = fresh()
        st = training.DataParallelStep(m, lr=0.01, world_size=1, precision=16)
        for _ in range(4):
            st.step(x, y)
        torch.cuda.synchronize()
        assert st._plans
        ends.append(torch.cat([st.eng.arena.p, st.eng.arena.m]).clone())
        del st, m
    assert torch.equal(ends[0], ends[1])


def test_conv3x3_c64_halo_kernel(dev):
    """Halo-tile 64 -> 64 conv (layer1 forward / input gradient): plain, with residual, with the fused BatchNorm + ReLU input
    transform (+ emitted activation) and with the output statistics -- ragged tiles (maps that are not multiples of 8 x 16),
    a single partial tile, and the 64 x 64 maps of the benchmark."""
    from self_supervised import ops
    for (n, h, w) in [(2, 8, 16), (3, 12, 20), (1, 5, 7), (2, 64, 64), (5, 24, 24)]:
        g = torch.Generator().manual_seed(n * 100 + h)
        x = torch.randn(n, 64, h, w, generator=g, requires_grad=True)
        wt = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).requires_grad_()
        res = torch.randn(n, 64, h, w, generator=g)
        nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
        w_ohwi = ops.repack_oihw_to_ohwi(wt.detach().to(dev))
        y = F.conv2d(x, wt, None, 1, 1)
        got = ops.conv3x3_c64(nh(x), w_ohwi)
        assert rel_err(got.permute(0, 3, 1, 2), y) < 2e-5
        assert rel_err(got, ops.conv_fwd(nh(x), w_ohwi, None, None, None, False, 1, 1)) < 2e-6
        got_r = ops.conv3x3_c64(nh(x), w_ohwi, residual=nh(res))
        assert rel_err(got_r.permute(0, 3, 1, 2), y + res) < 2e-5
        # as the input gradient of the same layer
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        dx = ops.conv3x3_c64(nh(dy), ops.flip_transpose_weight(w_ohwi), residual=nh(res))
        assert rel_err(dx.permute(0, 3, 1, 2), x.grad + res) < 2e-5
        # fused input transform relu(bn(x)) + emitted activation + output statistics
        mean, invstd = (torch.randn(64, generator=g) * 0.2).to(dev), (torch.rand(64, generator=g) + 0.5).to(dev)
        gamma, beta = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
        t_want = ops.bn_apply_fwd(nh(x), mean, invstd, gamma, beta, None, True)
        rm1, rv1 = torch.zeros(64, device=dev), torch.ones(64, device=dev)
        rm2, rv2 = rm1.clone(), rv1.clone()
        z_want, m_want, i_want = ops.conv_fwd_stats(t_want, w_ohwi, 1e-5, 0.1, rm1, rv1, 1, 1)
        z, t, m, i = ops.conv3x3_c64(nh(x), w_ohwi, transform=(mean, invstd, gamma, beta), emit=True, stats=(1e-5, 0.1, rm2, rv2))
        assert torch.equal(t, t_want)                                  # same expression as bn_apply_fwd, bit for bit
        assert rel_err(z, z_want) < 2e-6
        assert rel_err(m, m_want) < 1e-5 and rel_err(i, i_want) < 1e-5
        assert rel_err(rm2, rm1) < 1e-5 and rel_err(rv2, rv1) < 1e-5
        z2 = ops.conv3x3_c64(nh(x), w_ohwi, transform=(mean, invstd, gamma, beta))
        assert torch.equal(z2, z)                                      # deterministic; emit / stats do not change the result


def test_conv3x3_c64_eval_form(dev, monkeypatch):
    """Inference form of the halo-tile conv (scale / shift / residual / ReLU epilogue) in both layouts and across them,
    against torch and against the implicit-GEMM kernels it replaces in the scoring trunk; then a whole patch-scoring
    forward with the kernel switched on and off."""
    from self_supervised import ops
    for (n, h, w) in [(130, 16, 16), (3, 12, 20), (1, 5, 7), (7, 8, 8), (2, 64, 64)]:
        g = torch.Generator().manual_seed(n * 7 + h)
        x = torch.randn(n, 64, h, w, generator=g)
        wt = torch.randn(64, 64, 3, 3, generator=g) / 24.0
        res = torch.randn(n, 64, h, w, generator=g)
        sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
        nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
        hw = lambda t: t.permute(2, 3, 0, 1).contiguous().to(dev)
        w_ohwi = ops.repack_oihw_to_ohwi(wt.to(dev))
        want = F.relu(F.conv2d(x, wt, None, 1, 1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
        scd, shd = sc.to(dev), sh.to(dev)
        a = ops.conv3x3_c64_eval(nh(x), w_ohwi, scd, shd, nh(res), True)
        assert rel_err(a.permute(0, 3, 1, 2), want) < 2e-5
        assert rel_err(a, ops.conv_fwd(nh(x), w_ohwi, scd, shd, nh(res), True, 1, 1)) < 2e-6
        b = ops.conv3x3_c64_eval(hw(x), w_ohwi, scd, shd, hw(res), True, True, True)
        assert torch.equal(b.permute(2, 0, 1, 3), a)                      # the layout changes addresses, not arithmetic
        assert rel_err(b, ops.conv_fwd_hwnc(hw(x), w_ohwi, scd, shd, hw(res), True, 1, 1)) < 2e-6
        c = ops.conv3x3_c64_eval(nh(x), w_ohwi, scd, shd, hw(res), True, False, True)
        assert torch.equal(c, b)
        c = ops.conv3x3_c64_eval(nh(x), w_ohwi, scd, shd, nh(res), True, False, True, False)     # the scoring trunk's last layer1 conv
        assert torch.equal(c, b)
        d = ops.conv3x3_c64_eval(hw(x), w_ohwi, None, None, None, False, True, False)
        assert rel_err(d.permute(0, 3, 1, 2), F.conv2d(x, wt, None, 1, 1)) < 2e-5
    from self_supervised.models import PeraNet
    torch.manual_seed(3)
    m = PeraNet().to(dev).eval()
    m.enable_patch_level_mode()
    img = torch.rand(1, 3, 256, 256, device=dev)
    monkeypatch.setenv("SSAD_C64_EVAL", "0")
    ref = m(img)["latent_space"].clone()
    monkeypatch.setenv("SSAD_C64_EVAL", "1")
    got = m(img)["latent_space"]
    assert got.shape == ref.shape and rel_err(got, ref) < 1e-5


def test_wgrad3x3_halo_kernel(dev, monkeypatch):
    """Halo-tile weight gradient (3x3 / stride 1 or 2 / pad 1, channels multiples of 64) against autograd and against the
    split-over-pixels kernel it replaces: ragged tiles, maps narrower than a tile, several (co, ci) blocks, many splits."""
    from self_supervised import _hip, ops
    assert _hip.lib().ssad_wgrad3x3_halo_ok(64, 128, 3, 3, 2, 1) == 2 and _hip.lib().ssad_wgrad3x3_halo_ok(64, 128, 3, 3, 1, 1) == 1
    # the last five: stride 2 (round 3) -- even and odd input sizes (the last output row / column then has no kx = 2 tap inside),
    # maps narrower than a tile, both tile shapes (output wider than 8 or not)
    for (n, h, w, cin, cout, s) in [(3, 8, 8, 64, 64, 1), (2, 12, 20, 64, 128, 1), (5, 2, 2, 128, 256, 1), (4, 16, 16, 128, 64, 1),
                                    (70, 9, 17, 64, 64, 1), (16, 64, 64, 64, 64, 1),
                                    (3, 16, 16, 64, 128, 2), (2, 9, 21, 64, 64, 2), (5, 4, 4, 128, 256, 2), (4, 64, 64, 64, 128, 2),
                                    (33, 7, 40, 64, 64, 2)]:
        g = torch.Generator().manual_seed(n * 31 + h)
        x = torch.randn(n, cin, h, w, generator=g)
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).requires_grad_()
        y = F.conv2d(x, wt, None, s, 1)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        nh = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
        dw = torch.empty(cout * 9 * cin, device=dev)
        monkeypatch.setenv("SSAD_WGRAD_HALO", "1")
        ops.conv_wgrad(nh(dy), nh(x), dw, 3, 3, s, 1)
        assert rel_err(dw.view(cout, 3, 3, cin).permute(0, 3, 1, 2), wt.grad) < 2e-5, (n, h, w, cin, cout, s)
        dw2 = torch.empty_like(dw)
        ops.conv_wgrad(nh(dy), nh(x), dw2, 3, 3, s, 1)
        assert torch.equal(dw, dw2)                                   # deterministic
        monkeypatch.setenv("SSAD_WGRAD_HALO", "0")
        dw3 = torch.empty_like(dw)
        ops.conv_wgrad(nh(dy), nh(x), dw3, 3, 3, s, 1)
        assert rel_err(dw, dw3) < 2e-5
        dwo = torch.zeros(cout * 9 * cin, device=dev)
        monkeypatch.setenv("SSAD_WGRAD_HALO", "1")
        ops.conv_wgrad(nh(dy), nh(x), dwo, 3, 3, s, 1, to_oihw=True)
        assert torch.equal(dwo.view(cout, cin, 3, 3), dw.view(cout, 3, 3, cin).permute(0, 3, 1, 2))


def test_relu_mask_kernels(dev):
    """Residual-block BatchNorm with the final ReLU's active set kept as a nibble mask: forward output and mask, the two
    backward passes and the masked residual of both dgrad kernels equal the saved-activation path bit for bit."""
    from self_supervised import ops
    g = torch.Generator().manual_seed(3)
    for (n, h, c) in [(3, 10, 64), (2, 7, 128), (40, 1, 512)]:
        z = (torch.randn(n, h, h, c, generator=g) * 1.5 + 0.2).to(dev)
        res = torch.randn(n, h, h, c, generator=g).to(dev)
        gamma, beta = (torch.rand(c, generator=g) + 0.5).to(dev), (torch.randn(c, generator=g) * 0.3).to(dev)
        mean, invstd = ops.bn_stats(z, c, 1e-5, 0.1, None, None)
        y0 = ops.bn_apply_fwd(z, mean, invstd, gamma, beta, res, True)
        y1, mask = ops.bn_apply_fwd_mask(z, mean, invstd, gamma, beta, res, True)
        assert torch.equal(y0, y1)
        bits = torch.stack([(mask.view(-1, 1) >> k) & 1 for k in range(4)], 1).view(-1).bool()
        assert torch.equal(bits, (y0 > 0).view(-1))
        dy = torch.randn(z.shape, generator=g).to(dev)
        db0, dg0 = torch.empty(c, device=dev), torch.empty(c, device=dev)
        ops.bn_bwd_reduce(dy, y0, z, mean, invstd, db0, dg0, c)
        dz0, dres0 = ops.bn_apply_bwd(dy, y0, z, mean, invstd, gamma, db0, dg0, True)
        db1, dg1 = torch.empty(c, device=dev), torch.empty(c, device=dev)
        dz1 = ops.bn_bwd_mask(dy, mask, z, mean, invstd, gamma, db1, dg1)
        assert torch.equal(db0, db1) and torch.equal(dg0, dg1) and torch.equal(dz0, dz1)
        # no ReLU (downsample BatchNorm) fed by a masked gradient == the same fed by the materialised dres
        db2, dg2 = torch.empty(c, device=dev), torch.empty(c, device=dev)
        ops.bn_bwd_reduce(dres0, None, z, mean, invstd, db2, dg2, c)
        dz2, _ = ops.bn_apply_bwd(dres0, None, z, mean, invstd, gamma, db2, dg2, False)
        db3, dg3 = torch.empty(c, device=dev), torch.empty(c, device=dev)
        dz3 = ops.bn_bwd_mask(dy, mask, z, mean, invstd, gamma, db3, dg3)
        assert torch.equal(db2, db3) and torch.equal(dz2, dz3)
        # masked residual in the input-gradient kernels
        w = (torch.randn(c, 3, 3, c, generator=g) / (9 * c) ** 0.5).to(dev)
        up = torch.randn(z.shape, generator=g).to(dev)
        a = ops.conv_dgrad(up, w, z.shape, 1, 1, dres0)
        b = ops.conv_dgrad(up, w, z.shape, 1, 1, dy, res_mask=mask)
        assert torch.equal(a, b)
        if c == 64:
            assert torch.equal(ops.conv3x3_c64(up, w, residual=dres0), ops.conv3x3_c64(up, w, residual=dy, res_mask=mask))


def test_stem_conv_with_statistics(dev):
    """conv1 + the train-mode statistics of bn1 from the accumulators == conv1 followed by the separate statistics pass."""
    from self_supervised import ops
    for (b, h, w) in [(3, 64, 64), (2, 70, 90), (5, 32, 32), (4, 256, 256)]:
        g = torch.Generator().manual_seed(b + h)
        img = torch.randn(b, 3, h, w, generator=g).to(dev)
        wk = ops.pack_stem_weight((torch.randn(64, 3, 7, 7, generator=g) / 12).to(dev))
        z0 = ops.stem_fwd(img, wk, None, None, relu=False)
        rm0, rv0 = torch.zeros(64, device=dev), torch.ones(64, device=dev)
        m0, i0 = ops.bn_stats(z0, 64, 1e-5, 0.1, rm0, rv0)
        rm1, rv1 = torch.zeros(64, device=dev), torch.ones(64, device=dev)
        z1, m1, i1 = ops.stem_fwd_stats(img, wk, 1e-5, 0.1, rm1, rv1)
        assert torch.equal(z0, z1)
        assert rel_err(m1, m0) < 1e-6 and rel_err(i1, i0) < 1e-6 and rel_err(rm1, rm0) < 1e-6 and rel_err(rv1, rv0) < 1e-6
        z2, m2, i2 = ops.stem_fwd_stats(img, wk, 1e-5, 0.1, None, None)
        assert torch.equal(m2, m1) and torch.equal(i2, i1)          # deterministic
