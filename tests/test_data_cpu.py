"""CPU tests of the host-side data path: cut-paste primitives against vectors from the reference, datasets on a
synthetic MVTec-shaped tree, metrics."""
import os
import random

import numpy as np
import pytest
import torch
from PIL import Image

from fake_mvtec import make_tree


def test_cutpaste_primitives_match_reference(golden):
    from self_supervised import dataset_generator as dg
    g = golden("cutpaste")
    img = Image.fromarray(g["base"], "RGB")
    random.seed(123)
    patch = dg.generate_patch(img, area_ratio=(0.03, 0.07), aspect_ratio=((0.3, 0.5), (1, 3.3)))
    assert np.array_equal(np.array(patch), g["patch"])
    random.seed(124)
    mask = dg.rect2poly(patch, regular=False, sides=8)
    assert np.array_equal(np.array(mask), g["mask_rgba"])
    coords = dg.check_valid_coordinates_by_container((256, 256), patch.size, current_coords=(250, 10), container_scaling_factor=1.75)
    assert tuple(coords) == tuple(g["coords"])
    assert np.array_equal(np.array(dg.paste_patch(img, patch, coords, mask)), g["pasted"])
    c = dg.Container((256, 256), 1.75)
    assert [c.center, c.dim, c.left, c.top, c.right, c.bottom, c.width, c.height] == list(g["container"])
    assert abs(dg.check_color_similarity(img.crop((0, 0, 40, 40)), patch) - float(g["color_sim"])) < 1e-9
    random.seed(125)
    avg = dg.generate_patch(img, area_ratio=(0.2, 0.5), colorized=True, color_type="average")
    assert list(avg.size) == list(g["avg_patch_size"]) and list(np.array(avg)[0, 0]) == list(g["avg_patch_color"])


def test_obj_mask_finds_the_object():
    from self_supervised.dataset_generator import obj_mask
    yy, xx = np.mgrid[0:128, 0:128]
    disk = ((yy - 64) ** 2 + (xx - 60) ** 2) < 40 ** 2
    img = np.where(disk[..., None], 180, 20).astype(np.uint8) * np.ones((1, 1, 3), np.uint8)
    m = np.array(obj_mask(Image.fromarray(img)).convert('1'))
    inter, union = (m & disk).sum(), (m | disk).sum()
    assert inter / union > 0.9


def test_datasets_on_synthetic_tree(tmp_path):
    from self_supervised import datasets as ds
    root = make_tree(str(tmp_path / "data"))
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    dm = ds.PretextTaskDatamodule("bottle", root + "bottle/", imsize=(96, 96), batch_size=4, min_dataset_length=12, seed=0)
    dm.setup()
    assert len(dm.train_dataset) >= 12 and len(dm.val_dataset) >= 12
    # quirk Q1: the train set is built from the validation split (2 of 10 files, duplicated)
    assert len(set(dm.train_dataset.images_filenames)) == 2 and len(set(dm.val_dataset.images_filenames)) == 8
    labels = []
    for i in range(12):
        x, y, orig = dm.train_dataset[i]
        assert tuple(x.shape) == (3, 96, 96) and x.dtype == torch.float32 and tuple(orig.shape) == (3, 96, 96)
        assert 0 <= y <= 3 and 0.0 <= float(orig.min()) and float(orig.max()) <= 1.0
        labels.append(y)
    assert len(set(labels)) >= 3
    pl = ds.PretextTaskDatamodule("carpet", root + "carpet/", imsize=(96, 96), batch_size=4, min_dataset_length=8,
                                  patch_localization=True, patch_size=32)
    pl.setup()
    x, y, orig = pl.train_dataset[0]
    assert tuple(x.shape) == (3, 32, 32) and tuple(orig.shape) == (3, 96, 96)
    mv = ds.MVTecDatamodule(root + "bottle/", imsize=(96, 96), batch_size=1)
    mv.setup()
    assert len(mv.test_dataset) == 6
    gts = [mv.test_dataset[i][1] for i in range(6)]
    assert tuple(gts[0].shape) == (1, 96, 96) and sum(float(g.sum()) > 0 for g in gts) == 3
    xb, gb, ob = next(iter(torch.utils.data.DataLoader(mv.test_dataset, batch_size=2)))
    assert tuple(xb.shape) == (2, 3, 96, 96)


def test_loader_workers_from_a_fork_server(tmp_path, monkeypatch):
    """datasets._worker_context: plain fork while no GPU is initialised (the reference's DataLoader default), a fork server once it is
    (or on request) -- the datasets pickle, the workers deliver the same batches as the in-process loader."""
    from self_supervised import datasets as ds
    monkeypatch.delenv("SSAD_LOADER_CONTEXT", raising=False)
    assert ds._worker_context().get_start_method() == ("forkserver" if torch.cuda.is_initialized() else "fork")
    monkeypatch.setenv("SSAD_LOADER_CONTEXT", "forkserver")
    assert ds._worker_context().get_start_method() == "forkserver"
    root = make_tree(str(tmp_path / "data"))
    monkeypatch.setattr(ds._DataModule, "num_workers", 2)
    mv = ds.MVTecDatamodule(root + "bottle/", imsize=(96, 96), batch_size=1)
    mv.setup("predict")
    got = [b for b in mv.predict_dataloader()]
    assert len(got) == 6
    monkeypatch.setattr(ds._DataModule, "num_workers", 0)
    want = [b for b in mv.predict_dataloader()]
    assert all(torch.equal(u, v) for a, b in zip(got, want) for u, v in zip(a, b))


def test_unguarded_driver_script_with_fork_server_workers(tmp_path):
    """The reference's drivers call training(...) at module level (src/test_training.py has no __main__ guard).  With fork-server
    workers such a script must still run ONCE: the loader hides __main__ from the worker bootstrap."""
    import subprocess
    import sys
    root = make_tree(str(tmp_path / "data"))
    script = tmp_path / "driver.py"
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "self-supervised-anomaly-detection_amd")
    script.write_text(
        "import sys\n"
        f"sys.path.insert(0, {pkg!r})\n"
        "from self_supervised import datasets as ds\n"
        "print('DRIVER BODY RUNS', flush=True)\n"
        "ds._DataModule.num_workers = 2\n"
        f"mv = ds.MVTecDatamodule({root + 'bottle/'!r}, imsize=(96, 96), batch_size=1)\n"
        "mv.setup('predict')\n"
        "print('BATCHES', sum(1 for _ in mv.predict_dataloader()), flush=True)\n")
    env = dict(os.environ, SSAD_LOADER_CONTEXT="forkserver")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:]
    assert p.stdout.count("DRIVER BODY RUNS") == 1 and "BATCHES 6" in p.stdout, p.stdout[-2000:]


def test_metrics(golden):
    from self_supervised import metrics as m
    g = golden("auroc")
    fpr, tpr, _ = m.compute_roc(torch.from_numpy(g["labels"]), torch.from_numpy(g["scores"]))
    assert abs(m.compute_auc(fpr, tpr) - float(g["auroc"])) < 1e-12
    # PRO on a case small enough for a brute-force statement
    rng = np.random.RandomState(3)
    maps = rng.rand(2, 12, 12)
    gts = np.zeros((2, 12, 12), int); gts[0, 2:5, 2:6] = 1; gts[0, 8:10, 8:11] = 1; gts[1, 5:7, 1:4] = 1
    fprs, pros = m.compute_pro(maps, gts)
    ok = gts == 0
    regions = [(0, slice(2, 5), slice(2, 6)), (0, slice(8, 10), slice(8, 11)), (1, slice(5, 7), slice(1, 4))]
    for t in np.sort(np.unique(maps))[::-1][::7]:
        pred = maps >= t
        fpr_t = (pred & ok).sum() / ok.sum()
        pro_t = np.mean([pred[i, ys, xs].mean() for i, ys, xs in regions])
        j = int(pred.sum())                       # scores are distinct: point j = after the j best pixels
        assert abs(fprs[j] - fpr_t) < 1e-6 and abs(pros[j] - pro_t) < 1e-9
    assert fprs[0] == 0 and pros[-1] == 1
    x = np.array([0, 0.1, 0.2, 0.5, 1.0]); y = np.array([0, 0.5, 0.6, 0.8, 1.0])
    assert abs(m.trapezoid(x, y) - np.trapezoid(y, x)) < 1e-12
    assert abs(m.trapezoid(x, y, x_max=0.3) - (0.025 + 0.055 + 0.5 * (0.6 + 0.6 + 0.2 / 3) * 0.1)) < 1e-12
    assert 0 < m.compute_aupro(fprs, pros, 0.3) <= 1
    s = torch.tensor([0.1, 0.4, 0.35, 0.8]); t = torch.tensor([0, 0, 1, 1])
    thr = m.best_f1_threshold(s, t)
    assert abs(thr - 0.35) < 1e-6 and abs(m.compute_f1(t, s, thr) - 0.8) < 1e-9
    assert abs(m.compute_iou(s, t, 0.5) - np.mean([2 / 3, 1 / 2])) < 1e-9


def test_metrics_match_reference(golden):
    """metrics.py / tools.Evaluator against the reference's own metrics.py:49-228 and tools.py:52-146 run on seeded maps
    (tests/golden/make_fixtures.py metrics): PRO curve point for point, AUPRO at 0.3 and 1.0, the clipped trapezoid with x_max on
    and off a curve point, ROC / AUC, the F1-optimal threshold, F1, IoU, and Evaluator.evaluate's pixel- and image-level scores."""
    from self_supervised import metrics as m, tools
    from self_supervised.constants import ModelOutputsContainer
    g = golden("metrics")
    for name in (str(c) for c in g["cases"]):
        maps, gts = torch.from_numpy(g[name + "_maps"]), torch.from_numpy(g[name + "_gts"]).float()
        fprs, pros = m.compute_pro(maps.squeeze(1).numpy(), gts.squeeze(1).numpy())
        assert fprs.shape == g[name + "_fprs"].shape
        assert np.abs(fprs - g[name + "_fprs"]).max() < 2e-7 and np.abs(pros - g[name + "_pros"]).max() < 1e-12
        assert abs(m.compute_aupro(fprs, pros, 0.3) - float(g[name + "_aupro03"])) < 1e-7
        assert abs(m.compute_aupro(fprs, pros, 1.0) - float(g[name + "_aupro_full"])) < 1e-7
        rf, rp = g[name + "_fprs"], g[name + "_pros"]              # the integrator alone, on the reference's own curve: exact
        want = g[name + "_trap"]
        for v, w in zip(g[name + "_trap_xs"], want[:3]):
            assert abs(m.trapezoid(rf, rp, x_max=float(v)) - w) < 1e-13
        assert abs(m.trapezoid(rf, rp) - want[3]) < 1e-13
        flat_s, flat_t = maps.flatten(), gts.flatten()
        fpr, tpr, _ = m.compute_roc(flat_t, flat_s)
        assert np.array_equal(fpr, g[name + "_roc_fpr"]) and np.array_equal(tpr, g[name + "_roc_tpr"])
        assert abs(m.compute_auc(fpr, tpr) - float(g[name + "_auc"])) < 1e-15
        thr = m.best_f1_threshold(flat_s, flat_t)
        assert thr == float(g[name + "_threshold"])
        assert abs(m.compute_f1(flat_t, flat_s, thr) - float(g[name + "_f1"])) < 2e-7
        box = ModelOutputsContainer()
        box.anomaly_maps, box.ground_truths = maps, gts
        box.y_true_binary_labels = (gts.flatten(1).sum(1) > 0).long()
        ev = tools.Evaluator(evaluation_metrics=['auroc', 'aupro', 'iou'])
        ev.evaluate(box, name, None, patch_level=True)
        got = np.array([ev.scores.auroc, ev.scores.aupro, ev.scores.iou])
        assert np.abs(got - g[name + "_eval_pixel"]).max() < 2e-7, (got, g[name + "_eval_pixel"])
    s, l = torch.from_numpy(g["image_scores"]), torch.from_numpy(g["image_labels"])
    box = ModelOutputsContainer()
    box.anomaly_maps, box.y_true_binary_labels = s, l
    ev = tools.Evaluator(evaluation_metrics=['auroc', 'f1-score'])
    ev.evaluate(box, "image", None)
    assert abs(ev.scores.auroc - float(g["image_auroc"])) < 1e-15
    assert ev._get_threshold(s, l) == float(g["image_threshold"])
    assert abs(ev.scores.f1_score - float(g["image_f1"])) < 2e-7
    # a class that occurs neither in the prediction nor in the target scores 0 (JaccardIndex's absent_score)
    assert m.compute_iou(torch.zeros(4), torch.zeros(4), 0.5) == 0.5


def _getitem_tree(tmp_path, g):
    from fake_mvtec import make_tree
    return make_tree(str(tmp_path / "dataset"), categories=tuple(str(c) for c in g["tree"]), n_train=4, n_test_good=1, n_test_bad=1, size=96)


def test_pretext_getitem_matches_reference(golden, tmp_path):
    """PretextTaskDataset.__getitem__ against the reference's own __getitem__ (src/self_supervised/datasets.py:209-394),
    run by tests/golden/make_fixtures.py on the same synthetic tree under the same python / numpy / torch seeds:
    bit-identical uint8 images and labels for object + texture categories, image- and patch-level, and identical RNG
    streams afterwards (one extra or missing draw anywhere would shift them)."""
    import random
    from PIL import Image
    from self_supervised import datasets
    g = golden("getitem")
    root = _getitem_tree(tmp_path, g)
    n = int(g["n_samples"])
    for case in g["cases"]:
        subject, patch = str(case).split(":")
        patch = bool(int(patch))
        key = f"{subject}_{int(patch)}"
        names = np.array(sorted(os.path.join(root, subject, "train/good", f) for f in os.listdir(os.path.join(root, subject, "train/good"))))
        size, ps = (int(v) for v in g[key + "_size"])
        ds = datasets.PretextTaskDataset(subject, names, imsize=(size, size), transform=None, patch_localization=patch, patch_size=ps,
                                         dataset_root=root)
        assert np.array_equal(np.array(ds.fixed_segmentation.convert("1")), g[key + "_seg"]), key
        for s in range(n):
            random.seed(s); np.random.seed(s); torch.manual_seed(s)
            x, y, orig = ds[s % len(names)]
            assert y == int(g[key + "_y"][s]), (key, s)
            assert np.array_equal(np.array(x), g[key + "_x"][s]), (key, s, y)
            if s == 0:
                assert np.array_equal((orig * 255).round().byte().numpy(), g[key + "_orig0"])
        rng = np.array([random.random(), np.random.rand(), float(torch.rand(1))])
        assert np.array_equal(rng, g[key + "_rng"]), key
    assert set(np.concatenate([g[f"{c.split(':')[0]}_{c.split(':')[1]}_y"] for c in map(str, g["cases"])]).tolist()) == {0, 1, 2, 3}


def test_sample_defect_draws_like_getitem(golden, tmp_path):
    """The GPU pipeline's host-side sampler (augment.sample_defect) consumes the three RNG streams exactly as
    __getitem__ does -- i.e. as the reference does, by the test above -- and draws the same label: after one call from the
    same seeds the python / numpy / torch generators are in the same state as after the PIL path."""
    import random
    from PIL import Image
    from self_supervised import augment, datasets
    g = golden("getitem")
    root = _getitem_tree(tmp_path, g)
    checked = 0
    for case in g["cases"]:
        subject, patch = str(case).split(":")
        patch = bool(int(patch))
        names = np.array(sorted(os.path.join(root, subject, "train/good", f) for f in os.listdir(os.path.join(root, subject, "train/good"))))
        size, ps = (int(v) for v in g[f"{subject}_{int(patch)}_size"])
        ds = datasets.PretextTaskDataset(subject, names, imsize=(size, size), transform=None, patch_localization=patch, patch_size=ps,
                                         dataset_root=root)
        cuts = np.stack([np.asarray(c) for c in ds.images_for_cut]) if subject == "carpet" else None
        for s in range(int(g["n_samples"])):
            img = np.asarray(Image.open(names[s % len(names)]).resize((size, size)).convert("RGB"))
            if subject == "screw":             # non-fixed object: the mask comes from the sample itself
                from self_supervised.dataset_generator import obj_mask
                seg = np.asarray(obj_mask(Image.fromarray(img)).convert("1"))
            else:
                seg = np.asarray(ds.fixed_segmentation.convert("1"))
            random.seed(s); np.random.seed(s); torch.manual_seed(s)
            _, y, _ = ds[s % len(names)]
            want = (random.random(), np.random.rand(), float(torch.rand(1)))
            random.seed(s); np.random.seed(s); torch.manual_seed(s)
            rec, _ = augment.sample_defect(subject, img, seg, cuts, patch, ps)
            got = (random.random(), np.random.rand(), float(torch.rand(1)))
            assert int(rec["label"]) == y, (subject, patch, s)
            assert got == want, (subject, patch, s, y)
            checked += 1
    assert checked == 12 * len(g["cases"])


def test_cable_slic_presegmentation(tmp_path):
    """'cable' (datasets.py:201-206): SLIC super-pixels + mean colours before the object mask.  Properties of the restated SLIC
    (its equality with scikit-image is test_skimage_restatements_match_the_library): a handful of labels starting at 1, every
    label 4-connected, two flat colour halves separated exactly, determinism; and the dataset builds its fixed mask through it."""
    from scipy import ndimage
    from self_supervised import dataset_generator as dg, datasets
    img = np.zeros((64, 64, 3), np.uint8)
    img[:, :32] = (200, 40, 40)
    img[:, 32:] = (30, 60, 210)
    seg = dg.slic_superpixels(img, n_segments=5, sigma=2)
    assert seg.min() == 1 and 2 <= seg.max() <= 12
    for s in np.unique(seg):
        assert ndimage.label(seg == s)[1] == 1, "a super-pixel must be connected"
    left, right = set(np.unique(seg[:, :28]).tolist()), set(np.unique(seg[:, 36:]).tolist())
    assert not (left & right), "super-pixels must not straddle the colour edge"
    assert np.array_equal(seg, dg.slic_superpixels(img, n_segments=5, sigma=2))
    avg = dg.label_mean_rgb(seg, img)
    assert avg.shape == img.shape and np.abs(avg[:, :24].astype(int) - (200, 40, 40)).max() <= 20
    root = make_tree(str(tmp_path / "data"), categories=("cable", "carpet"), n_train=3, n_test_good=1, n_test_bad=1, size=96)
    names = np.array(sorted(os.path.join(root, "cable", "train/good", f) for f in os.listdir(os.path.join(root, "cable", "train/good"))))
    ds = datasets.PretextTaskDataset("cable", names, imsize=(64, 64), transform=None, dataset_root=root)
    m = np.array(ds.fixed_segmentation.convert("1"))
    assert m.shape == (64, 64) and 0 < m.sum() < 64 * 64
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    x, y, o = ds[0]
    assert x.size == (64, 64) and y in (0, 1, 2, 3)


def test_skimage_restatements_match_the_library(golden):
    """scikit-image is a dependency of the reference (feature.canny inside obj_mask, dataset_generator.py:27-39; slic +
    label2rgb for 'cable', datasets.py:203-204) but not of this package.  tests/golden/skimage.npz holds what scikit-image
    0.18.3 -- and the REFERENCE's own obj_mask running on it -- return for eight synthetic images (generated with the real
    library by tests/golden/make_skimage_fixtures.py in the build container's conda interpreter).  The restatements must
    reproduce them exactly: edge maps, object masks, super-pixel labels (0.18 pre-processing and the >= 0.19 one assembled
    from the library's own building blocks), the mean-colour image."""
    from PIL import Image
    from self_supervised import dataset_generator as dg
    d = golden("skimage")
    assert "scikit-image 0.18.3" in list(d["versions"])
    for i in range(int(d["n"])):
        im = d[f"img{i}"]
        gray = np.array(Image.fromarray(im).convert("L"))
        assert np.array_equal(gray, d[f"gray{i}"])
        assert np.array_equal(dg._canny(gray, 1.5, 5, 15), d[f"canny{i}"])
        assert np.array_equal(np.array(dg.obj_mask(Image.fromarray(im)).convert("1")), d[f"mask{i}"])
        assert np.abs(dg._rgb2lab(im.astype(np.float64) / 255.0) - d[f"lab{i}"]).max() < 1e-9
        assert np.array_equal(dg.slic_superpixels(im, 5, 2, rescale=False), d[f"slic18_{i}"])
        seg = dg.slic_superpixels(im, 5, 2)
        assert np.array_equal(seg, d[f"slic19_{i}"])
        assert np.array_equal(dg.label_mean_rgb(seg, im), d[f"avg19_{i}"])


def test_canny_restatement_properties():
    """dataset_generator._canny follows skimage's documented pipeline (pinned by the test above): on a disc it returns a thin,
    closed, one-component contour at the disc's radius, nothing on a flat image, and nothing on the one-pixel border."""
    from self_supervised import dataset_generator as dg
    from scipy import ndimage
    yy, xx = np.mgrid[0:96, 0:96]
    r = np.hypot(yy - 48, xx - 48)
    img = np.where(r < 30, 200, 30).astype(np.uint8)
    e = dg._canny(img, 1.5, 5, 15)
    assert e.dtype == bool and e.any()
    assert not e[0].any() and not e[-1].any() and not e[:, 0].any() and not e[:, -1].any()
    assert np.abs(r[e] - 30).max() < 1.6                                   # on the contour
    assert ndimage.label(e, structure=np.ones((3, 3), int))[1] == 1          # one closed curve ...
    filled = ndimage.binary_fill_holes(e)
    assert abs(int(filled.sum()) - np.pi * 30 ** 2) < 0.06 * np.pi * 30 ** 2  # ... that encloses the disc
    assert e.sum() < 2.2 * 2 * np.pi * 30                                   # thin (non-maximum suppression)
    assert not dg._canny(np.full((64, 64), 77, np.uint8), 1.5, 5, 15).any()
    # hysteresis: a faint edge (below `high`) survives only when connected to a strong one
    faint = np.full((64, 64), 100, np.uint8)
    faint[:, 32:] = 105
    assert not dg._canny(faint, 1.5, 5, 15).any()
    faint[:, 32:] = (160 - 55 * np.arange(64) / 63).astype(np.uint8)[:, None]   # the same edge, fading from strong to faint
    assert dg._canny(faint, 1.5, 5, 15)[56:62, 30:35].any()


def test_glibc_hypot_restatement_matches_numpy():
    """csrc/objmask.hip restates np.hypot (= this image's libm: glibc 2.35, sqrt + one correction step, NOT correctly rounded in
    ~0.2 % of inputs) operation by operation; the same statement in numpy must agree with np.hypot bit for bit, or the device-side
    Canny cannot be expected to match the host one."""
    def hyp(x, y):
        x, y = np.abs(x), np.abs(y)
        ax, ay = np.where(x < y, y, x), np.where(x < y, x, y)
        h = np.sqrt(ax * ax + ay * ay)
        d1, d2 = h - ay, h - ax
        first = h <= 2.0 * ay
        t1 = np.where(first, ax * (2.0 * d1 - ax), 2.0 * d2 * (ax - 2.0 * ay))
        t2 = np.where(first, (d1 - 2.0 * (ax - ay)) * d1, (4.0 * d2 - ay) * ay + d2 * d2)
        with np.errstate(all="ignore"):
            r = h - (t1 + t2) / (2.0 * h)
            return np.where(ax >= ay / 2.0 ** -54, ax + ay, r)
    rng = np.random.RandomState(0)
    for scale in (1.0, 1e-3, 10.0):
        a = (rng.rand(400000) - 0.5) * 2 * scale * 10 ** rng.uniform(-5, 0, 400000)
        b = (rng.rand(400000) - 0.5) * 2 * scale * 10 ** rng.uniform(-5, 0, 400000)
        assert np.array_equal(np.hypot(a, b), hyp(a, b))
    z = np.array([0.0, 0.0, 3.0, -0.0]); w = np.array([0.0, 2.0, 4.0, -0.0])
    assert np.array_equal(np.hypot(z, w), hyp(z, w))
