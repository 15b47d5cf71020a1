"""Pins the CPU oracle against vectors emitted by the reference itself (tests/golden/make_fixtures.py)."""
import numpy as np
import torch

from oracle import weights as ow, scoring as osc
from oracle.peranet import OraclePeraNet, train_step, make_optimizer

RTOL, ATOL = 1e-5, 1e-5


def _model(sd, train=False):
    m = OraclePeraNet()
    m.load_state_dict(sd, strict=True)
    m.train(train)
    return m


def test_extract_patches_matches_reference(golden):
    g = golden("patches")
    a = torch.arange(2 * 3 * 48 * 40, dtype=torch.float32).reshape(2, 3, 48, 40)
    assert np.array_equal(osc.extract_patches(a, 32, 8).numpy().astype(np.int32), g["small"])
    big = torch.arange(3 * 256 * 256, dtype=torch.float32).reshape(1, 3, 256, 256)
    p = osc.extract_patches(big, 32, 8)
    assert tuple(p.shape) == tuple(g["big_shape"])
    for i, s in enumerate(g["big_sel"]):
        got = p[0, s, :, [0, 0, 31, 31], [0, 31, 0, 31]].numpy().astype(np.int32)
        assert np.array_equal(got, g["big_corners"][i])


def test_forward_image_level(golden, seeded_sd):
    g = golden("forward")
    m = _model(seeded_sd)
    with torch.no_grad():
        for key, x in (("img", ow.synthetic_images(2, 256, seed=1234)), ("c1", ow.synthetic_images(8, 64, seed=77)),
                       ("up", ow.synthetic_images(4, 32, seed=78)), ("up48", ow.synthetic_images(2, 48, seed=79))):
            o = m(x)
            np.testing.assert_allclose(o["classifier"].numpy(), g[key + "_logits"], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(o["latent_space"].numpy(), g[key + "_emb"], rtol=RTOL, atol=ATOL)


def test_forward_patch_level(golden, seeded_sd):
    g = golden("forward")
    m = _model(seeded_sd)
    m.patch_level = True
    with torch.no_grad():
        x = ow.synthetic_images(2, 64, seed=99)[:, :, :, :48].contiguous()
        o = m(x)
        assert [m.batch, m.num_patches] == list(g["psmall_bp"])
        np.testing.assert_allclose(o["latent_space"].numpy(), g["psmall_emb"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(o["classifier"].numpy(), g["psmall_logits"], rtol=RTOL, atol=ATOL)
        o = m(ow.synthetic_images(1, 256, seed=4321))
        assert (m.batch, m.num_patches) == (1, 841)
        emb = o["latent_space"].numpy()
        np.testing.assert_allclose(emb[g["patch_rows"]], g["patch_emb_rows"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(emb.astype(np.float64).sum(1), g["patch_emb_rowsum"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(o["classifier"].numpy(), g["patch_logits"], rtol=RTOL, atol=ATOL)


def test_train_step_and_sgd(golden, seeded_sd):
    g = golden("train_step")
    m = _model(seeded_sd, train=True)
    x, y = ow.synthetic_images(8, 64, seed=55), ow.synthetic_labels(8, seed=56)
    loss, acc, _ = train_step(m, x, y)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-6)
    params = dict(m.named_parameters())
    for n, ref in zip(g["grad_names"], g["grad_norms"]):
        np.testing.assert_allclose(params[str(n)].grad.double().norm().item(), ref, rtol=1e-4)
    np.testing.assert_allclose(params["classifier.weight"].grad.numpy(), g["grad_classifier_weight"], rtol=1e-5, atol=1e-7)
    bufs = dict(m.named_buffers())
    np.testing.assert_allclose(bufs["feature_extractor.bn1.running_var"].numpy(), g["bn1_running_var"], rtol=1e-5)
    opt, sched = make_optimizer(m, 0.03, 10, "projection_train")
    assert sched is None
    opt.step()
    np.testing.assert_allclose(params["classifier.weight"].detach().numpy(), g["post_step_classifier_weight"], rtol=1e-6, atol=1e-8)
    opt.zero_grad()
    loss2, _, _ = train_step(m, x, y)
    loss2.backward()
    opt.step()
    np.testing.assert_allclose(loss2.item(), g["loss2"], rtol=1e-5)
    np.testing.assert_allclose(params["classifier.weight"].detach().numpy(), g["post_step2_classifier_weight"], rtol=1e-5, atol=1e-7)


def test_detector_matches_reference(golden, seeded_sd):
    g = golden("detector")
    bank, qs = ow.synthetic_bank(588, 512, seed=2), ow.synthetic_bank(300, 512, seed=3)
    mean, _, _ = osc.cosine_knn_mean(bank.numpy(), qs.numpy(), 3)
    np.testing.assert_allclose(mean, g["kernel_scores"], rtol=0, atol=2e-6)
    # seeded split path: same np.random state as the fixture
    np.random.seed(11)
    d = osc.OracleAnomalyDetector()
    d.fit(bank.numpy())
    np.testing.assert_allclose(d.threshold, g["img_threshold"], atol=2e-6)
    np.testing.assert_allclose(d.predict(qs[:17].numpy()).numpy(), g["img_scores"], atol=2e-6)
    # patch-level: bank = 841 embeddings of the seed-4321 image, queries = 2 seed-2468 images
    m = _model(seeded_sd)
    m.patch_level = True
    with torch.no_grad():
        bank_src = m(ow.synthetic_images(1, 256, seed=4321))["latent_space"].numpy()
        q = m(ow.synthetic_images(2, 256, seed=2468))["latent_space"].numpy()
    np.random.seed(7)
    d = osc.OracleAnomalyDetector(patch_level=True, batch=2, num_patches=841)
    d.fit(bank_src)
    assert d.bank.shape[0] == int(g["bank_rows"])
    s = d.predict(q)
    assert tuple(s.shape) == (2, 1, 29, 29)
    np.testing.assert_allclose(s.numpy(), g["scores"], atol=1e-5)
    np.testing.assert_allclose(d.threshold, g["threshold"], atol=1e-5)


def test_upsample_restatement(golden):
    g = golden("upsample")
    maps = torch.from_numpy(g["maps"])
    k = osc.gaussian_kernel1d(7)
    assert abs(k.sum().item() - 1) < 1e-6 and torch.allclose(k, k.flip(0))
    np.testing.assert_allclose(k.numpy(), g["kernel1d"], rtol=1e-6)
    np.testing.assert_allclose(osc.gaussian_blur(maps, 7).numpy(), g["blurred"], rtol=1e-6, atol=1e-7)
    up = osc.upsample(maps, 256)
    np.testing.assert_allclose(up.numpy(), g["up256"], rtol=1e-6, atol=1e-7)
    # independent statement of the bilinear rule on one map
    blurred = torch.relu(osc.gaussian_blur(maps[:1], 7))[0, 0].numpy()
    np.testing.assert_allclose(osc.bilinear_loops(blurred, 64), g["up64"][0, 0], rtol=1e-5, atol=1e-6)


def test_auroc_case(golden):
    from sklearn.metrics import roc_curve, auc
    g = golden("auroc")
    fpr, tpr, _ = roc_curve(g["labels"], g["scores"])
    assert abs(auc(fpr, tpr) - float(g["auroc"])) < 1e-12


def test_gradcam_oracle_matches_reference(golden, seeded_sd):
    """oracle/gradcam.py against the reference's GradCam outputs (src/self_supervised/gradcam.py:25-48)."""
    from oracle.gradcam import gradcam
    from oracle.peranet import OraclePeraNet
    g = golden("gradcam")
    m = OraclePeraNet(); m.load_state_dict(seeded_sd)
    x = ow.synthetic_images(2, 64, seed=301)
    for key, xx, c in [("cam64_auto", x[0:1], None), ("cam64_c1", x[1:2], 1), ("cam64_c2", x[1:2], 2),
                       ("cam32_c1", ow.synthetic_images(1, 32, seed=303), 1)]:
        np.testing.assert_allclose(gradcam(m, xx, c).numpy(), g[key], atol=1e-6, rtol=0, err_msg=key)
