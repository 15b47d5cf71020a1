"""Synthetic-defect augmentation: host sampler (CPU) and the HIP batch kernel against PIL (GPU)."""
import random

import numpy as np
import pytest
import torch
from PIL import Image, ImageDraw, ImageEnhance


def _image(seed=0, size=128):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    img = np.stack([120 + 80 * np.sin(xx / 9.0), 100 + 60 * np.cos(yy / 7.0), 90 + 0.5 * xx], -1) + rng.randint(-10, 10, (size, size, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def _mask(size=128):
    yy, xx = np.mgrid[0:size, 0:size]
    return ((yy - size / 2) ** 2 + (xx - size / 2) ** 2) < (size * 0.42) ** 2


def test_sampler_records_are_well_formed():
    from self_supervised import augment
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    img, seg = _image(), _mask()
    seen = set()
    for subject, pl in (("bottle", False), ("hazelnut", False), ("carpet", True), ("bottle", True)):
        for _ in range(60):
            rec, (h, w) = augment.sample_defect(subject, img, seg if subject != "carpet" else np.ones_like(seg),
                                                cuts_u8=np.stack([_image(5), _image(6)]), patch_localization=pl, patch_size=64)
            y = int(rec["label"])
            seen.add(y)
            assert 0 <= y <= 3 and (h, w) == ((64, 64) if pl else (128, 128))
            n = int(rec["jit_n"])
            assert n == 3 and sorted(rec["jit_order"][:n].tolist()) == [0, 1, 2] and np.all(np.abs(rec["jit_factor"] - 1) <= 0.1 + 1e-6)
            if pl:
                assert 0 <= rec["crop_left"] <= 64 and 0 <= rec["crop_top"] <= 64 and rec["cut_w"] == 64
            else:
                assert rec["aff_on"] == (subject != "hazelnut") and rec["cut_w"] == 128
            if y == 1:
                assert 4 <= rec["poly_n"] <= 8 and rec["patch_w"] >= 2 and rec["patch_h"] >= 2
                assert rec["patch_dst_left"] >= 0 and rec["patch_dst_top"] >= 0
                frac = rec["patch_w"] * rec["patch_h"] / float(h * w)
                lo, hi = (0.2, 0.5) if pl else (0.03, 0.07)
                assert lo * 0.8 <= frac <= hi * 1.05
            if y == 2:
                assert 2 <= rec["scar_n"] <= 5 and rec["scar_rw"] >= rec["scar_w"] * 0.7
            if y == 3:
                assert 2 <= rec["line_n"] <= 32 and rec["line_width"] in (1, 3)
            if subject == "carpet":
                assert rec["cut_index"] in (0, 1)
    assert seen == {0, 1, 2, 3}


def test_window_sum_helper_matches_the_numpy_warp():
    """ssad_affine_window_sum_u8 (host C loop of the library) == summing the window of pil_exact.affine_nearest's warp, exactly:
    rotations + scales as RandomAffine draws them, windows inside the image and hanging over its right / bottom edge, no affine."""
    from self_supervised import augment, pil_exact as px
    from self_supervised.tv_transforms import inverse_affine_matrix
    rng = np.random.RandomState(5)
    for size in (64, 96, 256):
        img = _image(size, size)
        for _ in range(12):
            ang, sc = rng.uniform(-3, 3), rng.uniform(1.05, 1.1)
            aff = inverse_affine_matrix((size * 0.5, size * 0.5), ang, (0, 0), sc)
            fix = px.affine_fix_coeffs(aff)
            warped = px.affine_nearest(img, (size, size), aff)
            for left, top, w, h in ((0, 0, size, size), (5, 9, 32, 32), (size - 20, size - 10, 32, 32), (size - 1, 0, 64, 3)):
                for cur, f in ((warped, fix), (img, None)):
                    want = cur[top:min(top + h, size), left:min(left + w, size)].reshape(-1, 3).astype(np.float64).sum(0)
                    got = augment._window_sum(img, f, left, top, w, h)
                    assert np.array_equal(got, want), (size, left, top, w, h, f is None)


def test_sampler_pool_is_deterministic_per_batch(monkeypatch):
    """Sampler workers: the records of batch b depend on (base seed, stage, epoch, b, rank) only -- not on how many workers draw
    them, in which order they finish or how they were started (forked with the sampler in memory / started from a fork server and
    handed the category through shared-memory files) -- and equal what the same seed gives in-process."""
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    from self_supervised import augment
    imgs = np.stack([_image(s, 64) for s in range(6)])
    aug = augment.GpuCutPaste("bottle", imgs, np.broadcast_to(_mask(64), (6, 64, 64)), device="cpu")
    assert aug.host.masks_unique.shape[0] == 1                      # one mask for the category, kept once
    key = "test-pool"
    augment._POOL_STATE[key] = aug
    batches = [np.array([0, 1, 2, 3]), np.array([4, 5, 0, 1]), np.array([2, 2, 3, 5])]
    seeds = [augment._batch_seed(3, 1, b, 0) for b in range(3)]
    try:
        results = []
        for nw in (1, 3):
            with ProcessPoolExecutor(nw, mp_context=multiprocessing.get_context("fork")) as pool:
                futs = [pool.submit(augment._pool_sample, key, s, b) for s, b in zip(seeds, batches)]
                results.append([f.result() for f in reversed(futs)][::-1])
        # the product's pool: fork-server workers that map the published arrays
        monkeypatch.setenv("SSAD_LOADER_CONTEXT", "forkserver")
        desc, d = aug.host.publish()
        try:
            pool = augment.sampler_pool(2)
            assert augment._POOL["ctx"].get_start_method() == "forkserver"
            futs = [pool.submit(augment._pool_sample, desc, s, b) for s, b in zip(seeds, batches)]
            results.append([f.result() for f in futs])
        finally:
            import shutil
            shutil.rmtree(d, ignore_errors=True)
            augment._shutdown_pool()
        assert results[0] == results[1] == results[2]
        for (raw, hw), s, b in zip(results[0], seeds, batches):
            random.seed(s); np.random.seed(s % 2 ** 32); torch.manual_seed(s)
            recs, hw2 = aug.sample(b)
            assert recs.tobytes() == raw and tuple(hw) == tuple(hw2)
        assert len({r[0] for r in results[0]}) == 3 and augment._batch_seed(3, 1, 0, 0) != augment._batch_seed(3, 1, 0, 1)
        # a second stage (another fit numbering its epochs from 0) has its own seeds; stage 0 keeps the round-3 values
        assert augment._batch_seed(3, 1, 0, 0, stage=1) != augment._batch_seed(3, 1, 0, 0)
        assert augment._batch_seed(3, 1, 0, 0, stage=0) == augment._batch_seed(3, 1, 0, 0)
    finally:
        augment._POOL_STATE.pop(key, None)


def test_loader_stage_counter(tmp_path):
    """tools.training fits twice on one datamodule and each fit numbers its epochs from 0: the loader notices the counter going
    back and moves to the next stage, so neither the shuffle nor a batch seed of the first fit comes back; per-image masks
    (screw-like categories) are deduplicated by content."""
    import os
    from fake_mvtec import make_tree
    from self_supervised import augment, datasets
    root = make_tree(str(tmp_path / "dataset"), categories=("bottle",), n_train=10, n_test_good=1, n_test_bad=1, size=96)
    names = np.array(sorted(os.path.join(root, "bottle", "train/good", f) for f in os.listdir(os.path.join(root, "bottle", "train/good"))))
    ds = datasets.PretextTaskDataset("bottle", np.tile(names, 4), imsize=(64, 64), transform=datasets._default_transform(),
                                     dataset_root=root)
    ld = augment.GpuPretextLoader(ds, 8, shuffle=True, drop_last=True, num_workers=0, base_seed=11, device="cpu")
    seen = []
    for fit in range(2):
        for epoch in range(2):
            ld.shard(1, 0, epoch)
            seen.append((np.concatenate(ld._batches()).tobytes(), augment._batch_seed(ld.base_seed, ld.epoch, 0, ld.rank, ld.stage)))
    assert ld.stage == 1
    assert len({a for a, _ in seen}) == 4 and len({b for _, b in seen}) == 4
    ld.set_epoch(1); ld.set_epoch(1)                    # a validation loader asked for the same epoch again (another fit)
    assert ld.stage == 3
    masks = np.stack([_mask(64), _mask(64), ~_mask(64)])
    smp = augment.DefectSampler("screw", np.stack([_image(s, 64) for s in range(3)]), masks)
    assert smp.masks_unique.shape[0] == 2 and list(smp.mask_index) == [0, 0, 1] and np.array_equal(smp.masks, masks)


def _pil_reference(subject, names, size, patch, ps, root):
    from self_supervised import datasets
    return datasets.PretextTaskDataset(subject, names, imsize=(size, size), transform=datasets._default_transform(),
                                       patch_localization=patch, patch_size=ps, dataset_root=root)


@pytest.mark.gpu
def test_gpu_augmentation_is_byte_identical_to_pil(tmp_path):
    """The whole GPU pipeline (csrc/augment.hip driven by augment.sample_defect) against PretextTaskDataset.__getitem__ --
    itself pinned to the reference's __getitem__ by tests/test_data_cpu.py -- from the same python / numpy / torch seeds:
    the normalised fp32 image, the label and the `original` tensor must be EQUAL, for an object and a texture category,
    image- and patch-level, at two sizes (RandomAffine, polygon paste incl. out-of-image source crops and brightness
    decorrelation, rotated scars, poly-lines of width 1 and 3, ColorJitter in every sampled order)."""
    import os
    from fake_mvtec import make_tree
    from self_supervised import augment
    dev = torch.device("cuda:0")
    root = make_tree(str(tmp_path / "dataset"), categories=("bottle", "carpet", "capsule", "screw", "cable"), n_train=4, n_test_good=1,
                     n_test_bad=1, size=160)
    labels = set()
    for subject, patch, size, ps, n in [("bottle", False, 64, 32, 40), ("bottle", True, 64, 32, 24), ("carpet", False, 64, 32, 40),
                                        ("carpet", True, 64, 32, 40), ("carpet", False, 128, 64, 24), ("carpet", True, 128, 64, 24),
                                        ("bottle", False, 128, 64, 16),
                                        # fixed pre-crops of datasets.py:244-249 (windows that leave a 256 / 128 px image: zero padding),
                                        # per-sample object masks (screw is a non-fixed object), the SLIC pre-segmented cable
                                        ("capsule", True, 256, 64, 12), ("capsule", True, 128, 32, 12), ("screw", True, 256, 64, 12),
                                        ("screw", False, 128, 64, 8), ("cable", False, 128, 64, 8)]:
        names = np.array(sorted(os.path.join(root, subject, "train/good", f) for f in os.listdir(os.path.join(root, subject, "train/good"))))
        ds = _pil_reference(subject, names, size, patch, ps, root)
        imgs = np.stack([np.asarray(Image.open(nm).resize((size, size)).convert("RGB")) for nm in names])
        if subject == "screw":                 # NON_FIXED_OBJECTS: a mask per image (datasets.py:232-233)
            from self_supervised.dataset_generator import obj_mask
            segs = np.stack([np.asarray(obj_mask(Image.fromarray(im)).convert("1")) for im in imgs])
        else:
            segs = np.broadcast_to(np.asarray(ds.fixed_segmentation.convert("1")), imgs.shape[:3])
        cuts = np.stack([np.asarray(c) for c in ds.images_for_cut]) if subject == "carpet" else None
        aug = augment.GpuCutPaste(subject, imgs, segs, cuts, patch, ps, device=dev)
        for s in range(n):
            i = s % len(names)
            random.seed(s); np.random.seed(s); torch.manual_seed(s)
            x_ref, y_ref, o_ref = ds[i]
            random.seed(s); np.random.seed(s); torch.manual_seed(s)
            x, y, o = aug([i])
            labels.add(int(y_ref))
            assert int(y[0]) == y_ref, (subject, patch, size, s)
            if not torch.equal(x[0].cpu(), x_ref):
                d = (x[0].cpu() != x_ref).any(0)
                raise AssertionError(f"{subject} patch={patch} size={size} seed={s} label={y_ref}: {int(d.sum())} pixels differ, "
                                     f"first at {np.argwhere(d.numpy())[:4].tolist()}")
            assert torch.equal(o[0].cpu(), o_ref)
    assert labels == {0, 1, 2, 3}


@pytest.mark.gpu
def test_gpu_kernel_rules_against_pillow():
    """Kernel-level: hand-built records for the corners the sampled cases rarely reach -- a polygon whose source crop leaves the
    cutting window (zero padding), a patch pasted across the image border, scars at every angle in [-45, 45] pasted over each
    other, long poly-lines of both widths running outside the image -- against Pillow itself, bit for bit."""
    from self_supervised import augment, _hip, pil_exact as px
    from self_supervised.dataset_generator import polygon_points
    dev = torch.device("cuda:0")
    img, cut = _image(3), _image(4)
    aug = augment.GpuCutPaste("carpet", img[None], np.ones((1, 128, 128), bool), cuts_u8=cut[None], device=dev)
    base = np.zeros((), augment.AUG_DTYPE)
    base["cut_w"], base["cut_h"] = 128, 128

    def run(r):
        params = torch.from_numpy(np.stack([r]).view(np.uint8).reshape(1, -1)).to(dev)
        work = torch.empty((1, 128, 128, 3), dtype=torch.uint8, device=dev)
        gm = torch.empty(1, device=dev); out = torch.empty((1, 3, 128, 128), device=dev)
        _hip.check(_hip.lib().ssad_cutpaste_augment(aug.images.data_ptr(), aug.cuts.data_ptr(), params.data_ptr(), work.data_ptr(),
                                                    gm.data_ptr(), out.data_ptr(), 1, 128, 128, 128, 128, aug._mean, aug._std,
                                                    _hip.stream()))
        return work.cpu()[0].numpy()

    rng = random.Random(7)
    for it in range(60):                                   # polygons
        random.seed(it)
        pw, ph = rng.randint(4, 70), rng.randint(4, 70)
        pts = polygon_points((pw, ph), sides=8)
        sl, st = rng.randint(60, 127), rng.randint(60, 127)            # source crop often leaves the 128 x 128 window
        dl, dt = rng.randint(-10, 110), rng.randint(-10, 110)          # and the paste box the image
        r = base.copy()
        r["label"], r["cut_index"] = 1, 0
        r["patch_src_left"], r["patch_src_top"], r["patch_w"], r["patch_h"] = sl, st, pw, ph
        r["patch_dst_left"], r["patch_dst_top"], r["poly_n"] = dl, dt, len(pts)
        r["poly_xy"][:2 * len(pts)] = np.asarray(pts, np.int32).ravel()
        if it % 3 == 0:
            r["patch_nbright"], r["patch_bright"] = 2, (rng.uniform(0.75, 0.9), rng.uniform(1.1, 1.15))
        mask = Image.new('RGBA', (pw, ph), (0, 0, 0, 0))
        ImageDraw.Draw(mask).polygon(pts, fill='white')
        patch = Image.fromarray(cut).crop((sl, st, sl + pw, st + ph))
        if it % 3 == 0:
            for f in r["patch_bright"]:
                patch = ImageEnhance.Brightness(patch).enhance(float(f))
        want = Image.fromarray(img).copy()
        want.paste(patch, (dl, dt), mask=mask)
        assert np.array_equal(run(r), np.asarray(want)), ("polygon", it)
    for angle in range(-45, 46):                           # scars
        sw, sh = rng.randint(2, 12), rng.randint(8, 40)
        r = base.copy()
        r["label"], r["scar_w"], r["scar_h"] = 2, sw, sh
        flat = angle % 2 == 0
        if flat:
            r["scar_flat"], r["scar_rgb"] = 1, (10, 200, 30)
            scar = Image.new('RGB', (sw, sh), (10, 200, 30))
        else:
            r["cut_index"], r["scar_src_left"], r["scar_src_top"] = 0, 40, 50
            scar = Image.fromarray(cut).crop((40, 50, 40 + sw, 50 + sh))
        scar = scar.convert('RGBA')
        s = scar.rotate(angle, expand=True)
        rw, rh, m = px.rotate_params(sw, sh, angle)
        assert (rw, rh) == s.size
        r["scar_rw"], r["scar_rh"], r["scar_n"] = rw, rh, 3
        if m is not None:
            r["scar_rot"], r["scar_fix"] = 1, px.affine_fix_coeffs(m)
        ats = [(rng.randint(-5, 100), rng.randint(-5, 100)) for _ in range(2)]
        ats.append((ats[0][0] + 3, ats[0][1] + 2))                       # overlaps the first copy
        r["scar_dst"][:6] = np.asarray(ats, np.int32).ravel()
        want = Image.fromarray(img).copy()
        for at in ats:
            want.paste(s, at, s)
        assert np.array_equal(run(r), np.asarray(want)), ("scar", angle)
    for it in range(60):                                   # poly-lines
        width = 1 if it % 2 else 3
        npts = rng.choice([2, 3, 6, 31, 32])
        pts = [(rng.uniform(-8, 136), rng.uniform(-8, 136)) for _ in range(npts)]
        if it % 5 == 0:
            pts[1] = (pts[0][0] + 0.3, pts[0][1] + 0.2)                  # a zero-length segment after truncation
        r = base.copy()
        ip = px.line_points_int(pts)
        r["label"], r["line_n"], r["line_rgb"], r["line_width"] = 3, npts, (192, 192, 192), width
        r["line_xy"][:2 * npts] = np.asarray(ip, np.int32).ravel()
        for k, ((x0, y0), (x1, y1)) in enumerate(zip(ip[:-1], ip[1:])):
            q = px.wide_line_quad(x0, y0, x1, y1, width) if width > 1 else None
            if q is not None:
                r["line_quad_ok"][k] = 1
                r["line_quad"][8 * k:8 * k + 8] = np.asarray(q, np.int32).ravel()
        want = Image.fromarray(img).copy()
        ImageDraw.Draw(want).line(pts, fill='silver', width=width)
        assert np.array_equal(run(r), np.asarray(want)), ("line", it, width)


@pytest.mark.gpu
def test_gpu_batches_feed_training(seeded_sd):
    from self_supervised import augment, training
    from self_supervised.models import PeraNet
    dev = torch.device("cuda:0")
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    imgs = np.stack([_image(s, 64) for s in range(12)])
    aug = augment.GpuCutPaste("bottle", imgs, np.broadcast_to(_mask(64), (12, 64, 64)), device=dev)
    x, y, orig = aug(np.arange(12))
    assert tuple(x.shape) == (12, 3, 64, 64) and x.dtype == torch.float32 and tuple(orig.shape) == (12, 3, 64, 64)
    assert y.dtype == torch.int64 and set(y.tolist()) <= {0, 1, 2, 3} and 0 <= orig.min() and orig.max() <= 1
    m = PeraNet(); m.load_state_dict(seeded_sd); m.to(dev).train(); m.unfreeze()
    step = training.DataParallelStep(m, lr=0.01, world_size=1)
    l0 = step.step(x, y)[0].item()
    for _ in range(5):
        l = step.step(x, y)[0].item()
    assert np.isfinite(l) and l < l0


@pytest.mark.gpu
def test_loader_with_sampler_workers(tmp_path):
    """GpuPretextLoader(num_workers > 0): the batches of an epoch are the same tensors whatever the number of sampler workers,
    differ between epochs, and come out in batch order."""
    import os
    from fake_mvtec import make_tree
    from self_supervised import augment, datasets
    root = make_tree(str(tmp_path / "dataset"), categories=("bottle",), n_train=10, n_test_good=1, n_test_bad=1, size=96)
    names = np.array(sorted(os.path.join(root, "bottle", "train/good", f) for f in os.listdir(os.path.join(root, "bottle", "train/good"))))
    ds = datasets.PretextTaskDataset("bottle", np.tile(names, 4), imsize=(64, 64), transform=datasets._default_transform(),
                                     dataset_root=root)
    out = {}
    for nw in (1, 3):
        ld = augment.GpuPretextLoader(ds, 8, shuffle=True, drop_last=True, num_workers=nw, base_seed=11)
        assert len(ld) == 5 and ld.speculate
        out[nw] = [[tuple(t.cpu() for t in b) for b in ld.shard(1, 0, e)] for e in (0, 1)]
        assert ld.spec_hits == 1                     # epoch 1 began with batches the pool had been handed during epoch 0
        if nw == 1:
            # a second fit on the same loader numbers its epochs from 0 again: new stage, new synthetic batches; and a
            # validation loader that is asked for the same epoch twice (two fits) does not repeat itself either
            again = [tuple(t.cpu() for t in b) for b in ld.shard(1, 0, 0)]
            assert ld.stage == 1 and not torch.equal(again[0][0], out[1][0][0][0])
            ld.set_epoch(0)
            third = [tuple(t.cpu() for t in b) for b in ld]
            assert ld.stage == 2 and not torch.equal(third[0][0], again[0][0])
        ld.close()
    # the cross-epoch hand-out changes when a batch is sampled, never what it is
    ld = augment.GpuPretextLoader(ds, 8, shuffle=True, drop_last=True, num_workers=2, base_seed=11)
    ld.speculate = False
    out[0] = [[tuple(t.cpu() for t in b) for b in ld.shard(1, 0, e)] for e in (0, 1)]
    assert ld.spec_hits == 0 and ld._spec is None
    ld.close()
    for e in (0, 1):
        assert len(out[1][e]) == 5
        for a, b, c in zip(out[1][e], out[3][e], out[0][e]):
            assert all(torch.equal(u, v) and torch.equal(u, w) for u, v, w in zip(a, b, c))
    assert not torch.equal(out[1][0][0][0], out[1][1][0][0])
    x, y, orig = out[1][0][0]
    assert tuple(x.shape) == (8, 3, 64, 64) and y.dtype == torch.int64 and tuple(orig.shape) == (8, 3, 64, 64)
