"""GPU end-to-end: tools.training (two stages) -> checkpoint -> tools.inference (patch level) -> upsample -> Evaluator,
on a synthetic MVTec-shaped tree; plus predict_step / validation_step against the oracle."""
import os

import numpy as np
import pytest
import torch

from fake_mvtec import make_tree

pytestmark = pytest.mark.gpu


def test_training_inference_evaluation_roundtrip(tmp_path):
    from self_supervised import tools, datasets
    datasets._DataModule.num_workers = 0
    root = make_tree(str(tmp_path / "data"), n_train=12, n_test_good=2, n_test_bad=2, size=96)
    out = str(tmp_path / "out") + "/"
    hist = tools.training(root + "bottle/", out, "bottle", imsize=(64, 64), batch_size=4, seed=0,
                          projection_training_params=(2, 0.03), fine_tune_params=(2, 0.005),
                          trainer_kwargs={"limit_train_batches": 3, "limit_val_batches": 2})
    assert os.path.exists(out + "best_model.ckpt")
    for stage in ("projection_train", "fine_tune"):
        assert len(hist[stage]["train"]["loss"]) == 2 and np.isfinite(hist[stage]["train"]["loss"]).all()
        assert len(hist[stage]["val"]["accuracy"]) == 2
    ck = torch.load(out + "best_model.ckpt", map_location="cpu", weights_only=False)
    assert len(ck["state_dict"]) == 153 and ck["hyper_parameters"]["stage"] == "fine_tune" and "memory_bank" in ck
    assert tuple(ck["state_dict"]["feature_extractor.conv1.weight"].shape) == (64, 3, 7, 7)
    assert ck["state_dict"]["feature_extractor.conv1.weight"].is_contiguous()
    np.random.seed(0)
    res = tools.inference(out + "best_model.ckpt", root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)
    assert tuple(res.anomaly_maps.shape) == (4, 1, 29, 29) and tuple(res.embedding_vectors.shape) == (4 * 841, 512)
    assert res.y_true_binary_labels.tolist().count(1) == 2
    up = tools.upsample(res.anomaly_maps, 256)
    assert tuple(up.shape) == (4, 1, 256, 256) and float(up.min()) >= 0
    res.anomaly_maps = up.cpu()
    ev = tools.Evaluator(evaluation_metrics=['auroc', 'aupro', 'iou'])
    ev.evaluate(res, "bottle", out, patch_level=True)
    assert 0.0 <= ev.scores.auroc <= 1.0 and 0.0 <= ev.scores.aupro <= 1.0 and 0.0 <= ev.scores.iou <= 1.0
    with pytest.raises(ValueError):
        tools.Evaluator(['f1-score']).evaluate(res, "bottle", None, patch_level=True)


def test_predict_and_validation_steps(seeded_sd):
    from self_supervised.models import PeraNet
    from oracle import weights as ow
    from oracle.peranet import OraclePeraNet
    dev = torch.device("cuda:0")
    m = PeraNet(); m.load_state_dict(seeded_sd); m.to(dev).eval()
    ref = OraclePeraNet(); ref.load_state_dict(seeded_sd); ref.eval()
    x = ow.synthetic_images(3, 64, seed=8)
    gts = torch.zeros(3, 1, 64, 64); gts[2, 0, 5, 5] = 1
    m.enable_mvtec_inference()
    out = m.predict_step((x.to(dev), gts.to(dev), x.to(dev)), 0)
    with torch.no_grad():
        ro = ref(x)
    assert out.y_true_binary_labels.tolist() == [0, 0, 1] and out.y_true_multiclass_labels.tolist() == [-1, -1, 4]
    assert torch.equal(out.y_hat.cpu(), ro["classifier"].argmax(1))
    assert (out.embedding_vectors.cpu() - ro["latent_space"]).abs().max() < 1e-4 * max(1.0, ro["latent_space"].abs().max().item())
    m.disable_mvtec_inference()
    out2 = m.predict_step((x.to(dev), torch.tensor([0, 2, 3]), x.to(dev)), 0)
    assert out2.y_true_binary_labels.tolist() == [0, 1, 1]
    y = ow.synthetic_labels(3, seed=9)
    m.train()
    v = m.validation_step((x.to(dev), y.to(dev), None), 0)
    want = torch.nn.functional.cross_entropy(ro["classifier"], y)
    assert abs(v["val_loss"].item() - want.item()) < 1e-4 and m.training


def test_predict_groups_batches_without_changing_results(seeded_sd):
    """Trainer.predict runs up to 16 images of consecutive bs=1 batches through one kernel sequence; the containers it
    returns (one per original batch) equal the ungrouped ones."""
    from oracle import weights as ow
    from self_supervised.models import PeraNet
    from self_supervised.trainer import Trainer
    m = PeraNet(); m.load_state_dict(seeded_sd); m.eval(); m.enable_patch_level_mode(); m.enable_mvtec_inference()
    imgs = ow.synthetic_images(5, 64, seed=81)
    gts = torch.zeros(5, 1, 64, 64); gts[1, 0, 3:9, 4:8] = 1; gts[4, 0, 30:40, 30:33] = 1
    loader = [(imgs[i:i + 1], gts[i:i + 1], imgs[i:i + 1] * 0.5) for i in range(4)] + [(imgs[3:5], gts[3:5], imgs[3:5])]
    t = Trainer(accelerator='auto', devices=1)
    t.predict_group = 1
    single = t.predict(m, dataloaders=loader)
    t.predict_group = 16
    grouped = t.predict(m, dataloaders=loader)
    assert len(single) == len(grouped) == 5
    for a, b in zip(single, grouped):
        a.to_cpu(); b.to_cpu()
        for f in ("original_data", "tensor_data", "y_true_binary_labels", "raw_predictions", "y_hat",
                  "y_true_multiclass_labels", "ground_truths", "embedding_vectors"):
            va, vb = getattr(a, f), getattr(b, f)
            assert va.shape == vb.shape, f
            assert torch.allclose(va.float(), vb.float(), atol=1e-6, rtol=0), f
    assert tuple(grouped[4].embedding_vectors.shape) == (2 * 25, 512)       # 64x64 image: 5x5 windows


def test_category_sweep(tmp_path):
    """tools.sweep over two synthetic categories: per-category training -> inference -> upsample -> Evaluator, one row of
    scores each plus the average row, csv export (BASELINE configs[4] shape of work, tiny)."""
    from self_supervised import tools, datasets
    datasets._DataModule.num_workers = 0
    root = make_tree(str(tmp_path / "data"), n_train=8, n_test_good=2, n_test_bad=2, size=96)
    out = str(tmp_path / "out") + "/"
    np.random.seed(0)
    df = tools.sweep(root, out, ["bottle", "carpet"], imsize=(64, 64), batch_size=4, seed=0,
                     projection_training_params=(1, 0.03), fine_tune_params=(1, 0.005),
                     trainer_kwargs={"limit_train_batches": 2, "limit_val_batches": 1}, tables_output=out + "tables/")
    assert list(df.index) == ["bottle", "carpet", "average"]
    assert {"auroc", "aupro", "iou"} <= set(df.columns)
    assert np.isfinite(df.loc["average"].values).all()
    np.testing.assert_allclose(df.loc["average", "auroc"], df.loc[["bottle", "carpet"], "auroc"].mean())
    assert os.path.exists(out + "tables/csv/patch_all_scores.csv")
    assert os.path.exists(out + "bottle/best_model.ckpt") and os.path.exists(out + "carpet/best_model.ckpt")


def test_inference_maps_match_the_oracle_end_to_end(tmp_path, seeded_sd):
    """tools.inference (checkpoint -> MVTec test images -> 841 patches each -> embeddings -> bank from the first training image,
    70 / 30 split from the global numpy RNG -> cosine 3-NN maps) against the same pipeline run with the CPU oracle on the same
    files, checkpoint and numpy seed: maps within 1e-4 (fp32), same bank split, same threshold."""
    from self_supervised import tools, datasets
    from oracle.peranet import OraclePeraNet
    from oracle import scoring as osc
    datasets._DataModule.num_workers = 0
    import shutil
    root = make_tree(str(tmp_path / "data"), categories=("bottle",), n_train=3, n_test_good=1, n_test_bad=1, size=96)
    # tools.inference takes the first batch of a SHUFFLED loader as the normality set (tools.py:379-381): make every training
    # image the same file so that the choice does not matter
    for k in (1, 2):
        shutil.copy(root + "bottle/train/good/000.png", root + f"bottle/train/good/{k:03d}.png")
    ck = str(tmp_path / "seeded.ckpt")
    torch.save({"state_dict": seeded_sd, "hyper_parameters": {}, "memory_bank": torch.tensor([])}, ck)
    np.random.seed(3)
    res = tools.inference(ck, root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)
    assert tuple(res.anomaly_maps.shape) == (2, 1, 29, 29)
    ref = OraclePeraNet(); ref.load_state_dict(seeded_sd); ref.eval(); ref.patch_level = True
    dm = datasets.MVTecDatamodule(root + "bottle/", batch_size=1)
    dm.setup()
    with torch.no_grad():
        test_emb = torch.cat([ref(x)["latent_space"] for x, _, _ in dm.predict_dataloader()])
        normality = ref(next(iter(dm.train_dataloader()))[0])["latent_space"]
    np.random.seed(3)
    det = osc.OracleAnomalyDetector(patch_level=True, batch=2, num_patches=841)
    det.fit(normality.numpy())
    want = det.predict(test_emb.numpy())
    assert (res.embedding_vectors - test_emb).abs().max().item() < 1e-4 * max(1.0, test_emb.abs().max().item())
    assert (res.anomaly_maps - want).abs().max().item() < 1e-4


def test_inference_bank_image_rides_with_the_test_images(tmp_path, seeded_sd, monkeypatch):
    """Round 6: tools.inference decides which training image becomes the normality bank BEFORE it predicts (the reference's two draws
    from torch's global generator, made early and in its order) and scores that image in front of the test images.  With DISTINCT
    training images the result must be the one of the DataLoader route (SSAD_FAST_PREDICT=0: Trainer.predict over the shuffled
    training loader, tools.py:374-381): same bank image -> same maps, and the global generator ends in the same state."""
    from self_supervised import tools, datasets
    datasets._DataModule.num_workers = 0
    root = make_tree(str(tmp_path / "data"), categories=("bottle",), n_train=7, n_test_good=2, n_test_bad=2, size=96)
    ck = str(tmp_path / "seeded.ckpt")
    torch.save({"state_dict": seeded_sd, "hyper_parameters": {}, "memory_bank": torch.tensor([])}, ck)
    outs = []
    for fast in ("1", "0"):
        monkeypatch.setenv("SSAD_FAST_PREDICT", fast)
        np.random.seed(3)
        torch.manual_seed(1234)
        res = tools.inference(ck, root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)
        outs.append((res, torch.get_rng_state().clone()))
    (a, sa), (b, sb) = outs
    assert torch.equal(sa, sb)
    assert torch.equal(a.anomaly_maps, b.anomaly_maps) and torch.equal(a.embedding_vectors, b.embedding_vectors)
    assert torch.equal(a.y_hat, b.y_hat) and torch.equal(a.ground_truths, b.ground_truths)
    # a different seed picks (with seven images: almost surely) another bank image: the maps move
    monkeypatch.setenv("SSAD_FAST_PREDICT", "1")
    np.random.seed(3)
    torch.manual_seed(99)
    c = tools.inference(ck, root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)
    np.random.seed(3)
    torch.manual_seed(1234)
    d = tools.inference(ck, root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)
    assert torch.equal(d.anomaly_maps, a.anomaly_maps)
    assert tuple(c.anomaly_maps.shape) == tuple(a.anomaly_maps.shape)


def test_category_sweep_all_fifteen(tmp_path):
    """BASELINE configs[4]'s shape of work on one GPU: tools.sweep over all fifteen MVTec-AD category names (objects with a fixed
    mask, non-fixed objects with per-image object masks, textures cutting defects from other images, the SLIC pre-segmented
    cable, capsule / screw with their fixed pre-crops) at 256 x 256 patch level, one short training each."""
    from self_supervised import tools, datasets, constants
    datasets._DataModule.num_workers = 0
    cats = list(constants.TEXTURES()) + [o for o in constants.OBJECTS() if o not in constants.TEXTURES()]   # 'tile' is in both lists
    assert len(cats) == 15
    root = make_tree(str(tmp_path / "data"), categories=tuple(cats), n_train=4, n_test_good=1, n_test_bad=1, size=96)
    out = str(tmp_path / "out") + "/"
    np.random.seed(0)
    df = tools.sweep(root, out, cats, imsize=(256, 256), batch_size=8, seed=0, projection_training_params=(1, 0.03),
                     fine_tune_params=(1, 0.005), trainer_kwargs={"limit_train_batches": 1, "limit_val_batches": 1},
                     tables_output=out + "tables/")
    assert list(df.index) == cats + ["average"] and {"auroc", "aupro", "iou"} <= set(df.columns)
    assert np.isfinite(df.values).all() and ((df["auroc"] >= 0) & (df["auroc"] <= 1)).all()
    assert all(os.path.exists(out + c + "/best_model.ckpt") for c in cats)


def test_gpu_auroc_matches_sklearn(golden):
    from sklearn.metrics import roc_auc_score
    from self_supervised import metrics as m
    dev = torch.device("cuda:0")
    g = golden("auroc")
    got = m.auroc_gpu(torch.from_numpy(g["labels"]).to(dev), torch.from_numpy(g["scores"]).to(dev))
    assert abs(got - float(g["auroc"])) < 1e-12
    gen = torch.Generator().manual_seed(1)
    for n, levels in ((1 << 20, 0), (300000, 17), (5000, 3), (2, 0)):
        y = (torch.rand(n, generator=gen) > 0.9)
        y[0], y[-1] = True, False
        s = torch.rand(n, generator=gen) + 0.3 * y
        if levels:                       # heavy ties: quantised scores
            s = (s * levels).floor() / levels
        want = roc_auc_score(y.numpy(), s.numpy())
        assert abs(m.auroc_gpu(y.to(dev), s.to(dev)) - want) < 1e-10, (n, levels)
    fpr, tpr, _ = m.compute_roc(y, s)
    assert abs(m.compute_auc(fpr, tpr) - want) < 1e-12
    # the hand-written radix sort on negative scores, signed zeros (equal as floats, different bit patterns) and a size that is
    # not a multiple of the 4 096-key tile
    n = 3 * 4096 + 1234
    y = torch.rand(n, generator=gen) > 0.7
    s = (torch.randn(n, generator=gen) * 2).round() / 2
    s[::5] = 0.0
    s[1::5] = -0.0
    s = s - 0.8 * y
    want = roc_auc_score(y.numpy(), s.numpy())
    assert abs(m.auroc_gpu(y.to(dev), s.to(dev)) - want) < 1e-10


def test_training_with_gpu_resident_pipeline(tmp_path):
    """tools.training(gpu_pipeline=True): batches are synthesised on the GPU (no DataLoader workers), precision=16 as the
    reference's Trainer asks; both stages run and the checkpoint loads."""
    from self_supervised import tools, datasets
    from self_supervised.models import PeraNet
    datasets._DataModule.num_workers = 0
    root = make_tree(str(tmp_path / "data"), n_train=12, n_test_good=2, n_test_bad=2, size=96)
    out = str(tmp_path / "out") + "/"
    hist = tools.training(root + "carpet/", out, "carpet", imsize=(64, 64), batch_size=4, seed=1, patch_localization=True,
                          patchsize=32, projection_training_params=(1, 0.03), fine_tune_params=(2, 0.005),
                          trainer_kwargs={"limit_val_batches": 2}, gpu_pipeline=True)
    assert len(hist["fine_tune"]["train"]["loss"]) == 2 and np.isfinite(hist["fine_tune"]["val"]["loss"]).all()
    m = PeraNet.load_from_checkpoint(out + "best_model.ckpt")
    assert m.stage == "fine_tune" and m.memory_bank.ndim in (1, 2)


def test_gpu_evaluation_metrics_match_the_host_functions(golden):
    """The PRO curve / AUPRO, the F1-optimal threshold, F1 and IoU of device-resident maps (csrc/auroc.hip: radix sort + scans)
    against metrics.compute_pro / best_f1_threshold / compute_f1 / compute_iou -- the host functions that tests/golden/metrics.npz
    pins to the reference's own metrics.py and Evaluator: thresholds EQUAL, curves point for point (fprs exact, pros to 1e-12: only
    the association of the fp64 running sum differs), Evaluator.evaluate on device maps == on host maps."""
    from self_supervised import metrics as m, tools
    from self_supervised.constants import ModelOutputsContainer
    dev = torch.device("cuda:0")
    g = golden("metrics")
    cases = [(torch.from_numpy(g[str(c) + "_maps"]), torch.from_numpy(g[str(c) + "_gts"]).float()) for c in g["cases"]]
    gen = torch.Generator().manual_seed(7)
    # a larger case: 24 maps of 128 x 128, heavy ties (scores on a grid of 97 levels), two defect-free images, touching regions
    maps = (torch.rand(24, 1, 128, 128, generator=gen) * 96).round() / 96
    gts = torch.zeros(24, 1, 128, 128)
    for i in range(22):
        y0, x0 = [int(v) for v in torch.randint(8, 90, (2,), generator=gen)]
        gts[i, 0, y0:y0 + 20 + i, x0:x0 + 14] = 1
        gts[i, 0, y0 + 21 + i:y0 + 30 + i, x0 + 14:x0 + 20] = 1          # touches the first block at a corner: one 8-connected region
        maps[i] += 0.5 * gts[i]
    cases.append((maps, gts))
    cases.append((torch.rand(3, 1, 64, 64, generator=gen), torch.zeros(3, 1, 64, 64)))          # no defect at all
    for maps, gts in cases:
        md = maps.to(dev)
        fprs, pros = m.compute_pro(maps.squeeze(1).numpy(), gts.squeeze(1).numpy())
        gf, gp = m.compute_pro_gpu(md.squeeze(1), gts.squeeze(1))
        assert gf.shape == fprs.shape and np.array_equal(gf, fprs) and np.abs(gp - pros).max() < 1e-12
        flat_s, flat_t = maps.flatten(), gts.flatten()
        thr = m.best_f1_threshold(flat_s, flat_t)
        assert m.best_f1_threshold_gpu(md.flatten(), flat_t) == thr
        assert abs(m.compute_f1_gpu(flat_t, md.flatten(), thr) - m.compute_f1(flat_t, flat_s, thr)) < 1e-15
        assert abs(m.compute_iou_gpu(md.flatten(), flat_t, thr) - m.compute_iou(flat_s, flat_t, thr)) < 1e-15
        host, devc = ModelOutputsContainer(), ModelOutputsContainer()
        host.anomaly_maps, host.ground_truths = maps, gts
        devc.anomaly_maps, devc.ground_truths = md, gts
        host.y_true_binary_labels = devc.y_true_binary_labels = (gts.flatten(1).sum(1) > 0).long()
        if gts.sum() > 0:
            a, b = tools.Evaluator(['auroc', 'aupro', 'iou']), tools.Evaluator(['auroc', 'aupro', 'iou'])
            a.evaluate(host, "x", None, patch_level=True)
            b.evaluate(devc, "x", None, patch_level=True)
            assert abs(a.scores.auroc - b.scores.auroc) < 1e-10 and abs(a.scores.aupro - b.scores.aupro) < 1e-10
            assert abs(a.scores.iou - b.scores.iou) < 1e-15
    # image level: a few hundred scores, the f1-score metric
    s, l = torch.from_numpy(g["image_scores"]), torch.from_numpy(g["image_labels"])
    host, devc = ModelOutputsContainer(), ModelOutputsContainer()
    host.anomaly_maps, host.y_true_binary_labels = s, l
    devc.anomaly_maps, devc.y_true_binary_labels = s.to(dev), l
    a, b = tools.Evaluator(['auroc', 'f1-score']), tools.Evaluator(['auroc', 'f1-score'])
    a.evaluate(host, "image", None)
    b.evaluate(devc, "image", None)
    assert abs(a.scores.auroc - b.scores.auroc) < 1e-12 and abs(a.scores.f1_score - b.scores.f1_score) < 1e-15
