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
            assert sorted(rec["jit_order"].tolist()) == [0, 1, 2] and np.all(np.abs(rec["jit_factor"] - 1) <= 0.1 + 1e-6)
            if pl:
                assert 0 <= rec["crop_left"] <= 64 and 0 <= rec["crop_top"] <= 64
            elif subject == "hazelnut":
                assert np.allclose(rec["aff"], (1, 0, 0, 0, 1, 0))
            else:
                assert not np.allclose(rec["aff"], (1, 0, 0, 0, 1, 0))
            if y == 1:
                assert 4 <= rec["poly_n"] <= 8 and rec["patch_w"] >= 2 and rec["patch_h"] >= 2
                assert rec["patch_dst_left"] >= 0 and rec["patch_dst_top"] >= 0
                frac = rec["patch_w"] * rec["patch_h"] / float(h * w)
                lo, hi = (0.2, 0.5) if pl else (0.03, 0.07)
                assert lo * 0.8 <= frac <= hi * 1.05
            if y == 2:
                assert 2 <= rec["scar_n"] <= 5 and rec["scar_rw"] >= rec["scar_w"] * 0.7
            if y == 3:
                assert 2 <= rec["line_n"] <= 32 and rec["line_width"] in (1.0, 3.0)
            if subject == "carpet":
                assert rec["cut_index"] in (0, 1)
    assert seen == {0, 1, 2, 3}


@pytest.mark.gpu
def test_kernel_identity_and_jitter():
    from self_supervised import augment, _hip
    import ctypes
    dev = torch.device("cuda:0")
    img = _image(2)
    aug = augment.GpuCutPaste("hazelnut", img[None], _mask()[None], device=dev)
    rec = np.zeros((), augment.AUG_DTYPE)
    rec["aff"], rec["cut_index"], rec["jit_order"], rec["jit_factor"] = (1, 0, 0, 0, 1, 0), -1, (0, 1, 2), (1, 1, 1)

    def run(r, h=128, w=128):
        params = torch.from_numpy(np.stack([r]).view(np.uint8).reshape(1, -1)).to(dev)
        work = torch.empty((1, h, w, 3), dtype=torch.uint8, device=dev)
        gm = torch.empty(1, device=dev); out = torch.empty((1, 3, h, w), device=dev)
        _hip.check(_hip.lib().ssad_cutpaste_augment(aug.images.data_ptr(), None, params.data_ptr(), work.data_ptr(), gm.data_ptr(),
                                                    out.data_ptr(), 1, 128, 128, h, w, aug._mean, aug._std, _hip.stream()))
        return out.cpu()[0], work.cpu()[0].numpy()

    mean, std = torch.tensor(augment.IMAGENET_MEAN).view(3, 1, 1), torch.tensor(augment.IMAGENET_STD).view(3, 1, 1)
    tt = lambda a: (torch.from_numpy(np.asarray(a)).permute(2, 0, 1).float() / 255 - mean) / std
    out, work = run(rec)
    assert np.array_equal(work, img) and torch.allclose(out, tt(img), atol=1e-6)
    # crop window
    r2 = rec.copy(); r2["crop_left"], r2["crop_top"] = 17, 40
    out, work = run(r2, 64, 64)
    assert np.array_equal(work, img[40:104, 17:81])
    # colour jitter vs PIL ImageEnhance in every order (uint8 blend, +-1 level)
    pil = Image.fromarray(img)
    for order in ((0, 1, 2), (2, 1, 0), (1, 0, 2), (1, 2, 0)):
        f = (1.08, 0.93, 1.07)
        r3 = rec.copy(); r3["jit_order"], r3["jit_factor"] = order, f
        want = pil
        for op in order:
            want = (ImageEnhance.Brightness, ImageEnhance.Contrast, ImageEnhance.Color)[op](want).enhance(f[op])
        out, _ = run(r3)
        got_u8 = ((out * std + mean) * 255).round()
        diff = (got_u8 - torch.from_numpy(np.asarray(want)).permute(2, 0, 1).float()).abs()
        assert diff.max() <= 2 and (diff > 1).float().mean() < 0.01, (order, diff.max())


@pytest.mark.gpu
def test_kernel_defects_against_pil():
    from self_supervised import augment, _hip
    dev = torch.device("cuda:0")
    img, cut = _image(3), _image(4)
    aug = augment.GpuCutPaste("carpet", img[None], np.ones((1, 128, 128), bool), cuts_u8=cut[None], device=dev)
    base = np.zeros((), augment.AUG_DTYPE)
    base["aff"], base["jit_order"], base["jit_factor"] = (1, 0, 0, 0, 1, 0), (0, 1, 2), (1, 1, 1)
    base["patch_bright"], base["scar_bright"] = (1, 1), (1, 1)

    def run(r):
        params = torch.from_numpy(np.stack([r]).view(np.uint8).reshape(1, -1)).to(dev)
        work = torch.empty((1, 128, 128, 3), dtype=torch.uint8, device=dev)
        gm = torch.empty(1, device=dev); out = torch.empty((1, 3, 128, 128), device=dev)
        _hip.check(_hip.lib().ssad_cutpaste_augment(aug.images.data_ptr(), aug.cuts.data_ptr(), params.data_ptr(), work.data_ptr(),
                                                    gm.data_ptr(), out.data_ptr(), 1, 128, 128, 128, 128, aug._mean, aug._std,
                                                    _hip.stream()))
        return work.cpu()[0].numpy()

    # polygon patch cut from another image
    pts = [(0, 30), (0, 8), (12, 0), (33, 0), (40, 11), (40, 25), (30, 36), (9, 36)]
    r = base.copy()
    r["label"], r["cut_index"] = 1, 0
    r["patch_src_left"], r["patch_src_top"], r["patch_w"], r["patch_h"] = 50, 60, 40, 36
    r["patch_dst_left"], r["patch_dst_top"], r["poly_n"] = 20, 70, len(pts)
    r["poly_xy"][:16] = np.asarray(pts, np.float32).ravel()
    got = run(r)
    mask = Image.new('L', (40, 36), 0)
    ImageDraw.Draw(mask).polygon(pts, fill=255)
    want = Image.fromarray(img).copy()
    want.paste(Image.fromarray(cut).crop((50, 60, 90, 96)), (20, 70), mask=mask)
    want = np.asarray(want)
    differ = np.any(got != want, axis=-1)
    assert differ.sum() <= 2 * (40 + 36) * 2          # only polygon-boundary pixels may differ (PIL's edge rule)
    inner = np.zeros((128, 128), bool); inner[80:96, 30:50] = True
    assert np.array_equal(got[inner], np.asarray(Image.fromarray(cut).crop((50, 60, 90, 96)))[10:26, 10:30].reshape(-1, 3))
    assert np.array_equal(got[:60], img[:60])
    # flat-colour rotated scars
    r = base.copy()
    r["label"], r["scar_w"], r["scar_h"], r["scar_flat"], r["scar_rgb"] = 2, 6, 30, 1, (10, 200, 30)
    a = np.deg2rad(30.0)
    r["scar_cos"], r["scar_sin"] = np.cos(a), np.sin(a)
    rw, rh = int(np.ceil(6 * np.cos(a) + 30 * np.sin(a))), int(np.ceil(6 * np.sin(a) + 30 * np.cos(a)))
    r["scar_rw"], r["scar_rh"], r["scar_n"] = rw, rh, 2
    r["scar_dst"][:4] = (10, 10, 80, 60)
    got = run(r)
    s = Image.new('RGBA', (6, 30), (10, 200, 30, 255)).rotate(30, expand=True)
    want = Image.fromarray(img).copy()
    for at in ((10, 10), (80, 60)):
        want.paste(s, at, s)
    gm, wm = np.any(got != img, -1), np.any(np.asarray(want) != img, -1)
    assert (gm & wm).sum() / (gm | wm).sum() > 0.8 and abs(int(gm.sum()) - 2 * 180) < 60
    assert np.all(got[gm] == np.array([10, 200, 30]))
    # poly-line
    r = base.copy()
    line = [(10, 20), (40, 35), (70, 30), (110, 90)]
    r["label"], r["line_n"], r["line_rgb"], r["line_width"] = 3, 4, (192, 192, 192), 3.0
    r["line_xy"][:8] = np.asarray(line, np.float32).ravel()
    got = run(r)
    want = Image.fromarray(img).copy()
    ImageDraw.Draw(want).line(line, fill='silver', width=3)
    gm, wm = np.any(got != img, -1), np.any(np.asarray(want) != img, -1)
    assert (gm & wm).sum() / (gm | wm).sum() > 0.7
    assert np.all(got[gm] == 192)


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
