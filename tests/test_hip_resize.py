"""Pillow's bicubic Image.resize on the device (csrc/resize.hip) and the loaders built on it (gpu_io): bit-exact against the
installed Pillow -- the reference resizes every file it opens on the host (src/self_supervised/datasets.py:68, :211-213)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _img(rng, h, w, c):
    yy, xx = np.mgrid[0:h, 0:w]
    base = (127 + 100 * np.sin(xx / 7.0 + c) * np.cos(yy / 11.0)).astype(np.int64)
    a = np.clip(base[..., None] + rng.randint(-40, 40, (h, w, c)), 0, 255).astype(np.uint8)
    return a if c == 3 else a[..., 0]


def test_resize_kernel_is_bit_exact():
    from self_supervised import ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(1)
    cases = [((700, 700), (256, 256)), ((1024, 1024), (256, 256)), ((900, 840), (256, 256)), ((256, 256), (64, 64)),
             ((100, 130), (256, 256)), ((257, 255), (256, 256)), ((840, 1000), (320, 200)), ((256, 300), (256, 256)),
             ((300, 256), (256, 256)), ((33, 47), (64, 64)), ((96, 96), (256, 256)), ((2048, 2048), (512, 512))]
    for (h, w), (oh, ow) in cases:
        for c in (1, 3):
            batch = np.stack([_img(rng, h, w, c) for _ in range(3)])
            if h * w < 100000:
                batch[1] = rng.randint(0, 256, batch[1].shape)           # dense noise: exercises the clipping at both ends
            dev_in = torch.from_numpy(batch if c == 3 else batch[..., None]).to(dev)
            got = ops.resize_bicubic_u8(dev_in, (ow, oh)).cpu().numpy()
            for i in range(3):
                want = np.asarray(Image.fromarray(batch[i]).resize((ow, oh)))
                assert np.array_equal(got[i] if c == 3 else got[i, ..., 0], want), ((h, w), (oh, ow), c, i)
    same = torch.zeros((1, 8, 8, 3), dtype=torch.uint8, device=dev)
    assert ops.resize_bicubic_u8(same, (8, 8)) is same                    # Image.resize to the same size is a copy


def test_file_loader_matches_pillow(tmp_path):
    """gpu_io.load_rgb_batch == np.asarray(Image.open(f).resize(size).convert('RGB')) for RGB, grey, palette and alpha files of
    mixed sizes (the last two through Pillow's own resize), in file order."""
    from self_supervised import gpu_io
    rng = np.random.RandomState(2)
    names = []
    for i, (h, w, mode) in enumerate([(300, 300, "RGB"), (300, 300, "L"), (512, 400, "RGB"), (300, 300, "RGB"), (256, 256, "RGB"),
                                      (256, 256, "L"), (128, 128, "P"), (200, 200, "RGBA"), (300, 300, "L")]):
        c = {"RGB": 3, "L": 1, "P": 3, "RGBA": 3}[mode]
        img = Image.fromarray(_img(rng, h, w, c))
        if mode == "P":
            img = img.convert("P")
        if mode == "RGBA":
            img = img.convert("RGBA")
            img.putalpha(Image.fromarray(_img(rng, h, w, 1)))
        fn = str(tmp_path / f"{i:02d}.png")
        img.save(fn)
        names.append(fn)
    got = gpu_io.load_rgb_batch(names, (256, 256), torch.device("cuda:0")).cpu().numpy()
    for i, fn in enumerate(names):
        want = np.asarray(Image.open(fn).resize((256, 256)).convert("RGB"))
        assert np.array_equal(got[i], want), fn


def test_loaders_on_native_size_files(tmp_path):
    """Files larger than the working size: the GPU-resident training loader holds the same uint8 images as the PIL path, and
    tools.inference's streamed predict returns the tensors MVTecDataset.__getitem__ gives (resize on the device, byte for byte)."""
    from fake_mvtec import make_tree
    from self_supervised import augment, datasets, tools
    from self_supervised.models import PeraNet
    root = make_tree(str(tmp_path / "dataset"), categories=("bottle",), n_train=5, n_test_good=2, n_test_bad=3, size=300)
    tr = sorted(os.path.join(root, "bottle", "train/good", f) for f in os.listdir(os.path.join(root, "bottle", "train/good")))
    ds = datasets.PretextTaskDataset("bottle", np.array(tr), imsize=(128, 128), transform=datasets._default_transform(), dataset_root=root)
    ld = augment.GpuPretextLoader(ds, 4, num_workers=0)
    want = np.stack([np.asarray(Image.open(n).resize((128, 128)).convert("RGB")) for n in tr])
    assert np.array_equal(ld.aug.images.cpu().numpy(), want) and np.array_equal(ld.aug.images_cpu, want)
    dm = datasets.MVTecDatamodule(root + "bottle/", imsize=(128, 128), batch_size=1)
    dm.setup()
    model = PeraNet().cuda().eval()
    model.enable_patch_level_mode(); model.enable_mvtec_inference()
    assert tools._fast_mvtec_ok(dm.test_dataset)
    out, emb_dev = tools._predict_mvtec_streamed(model, dm.test_dataset, torch.device("cuda:0"), list(range(len(dm.test_dataset))), group=2)
    for i in range(len(dm.test_dataset)):
        x, gt, orig = dm.test_dataset[i]
        assert torch.equal(out.tensor_data[i], x) and torch.equal(out.original_data[i], orig) and torch.equal(out.ground_truths[i], gt)
    assert torch.equal(out.embedding_vectors, emb_dev.cpu()) and out.embedding_vectors.shape[0] == 5 * model.num_patches
    assert out.y_true_binary_labels.tolist() == [1, 1, 1, 0, 0]              # test/broken sorts before test/good
    with torch.no_grad():
        ref = model(out.tensor_data[1:2].cuda())
    assert torch.equal(ref["latent_space"].cpu(), out.embedding_vectors[model.num_patches:2 * model.num_patches])


def _blob_image(rng, size, kind):
    yy, xx = np.mgrid[0:size, 0:size]
    bg = np.clip(40 + 8 * rng.randn(size, size, 3), 0, 255)
    img = bg.copy()
    if kind == "disc":
        sel = (yy - size * 0.52) ** 2 + (xx - size * 0.47) ** 2 < (size * 0.31) ** 2
        img[sel] = np.clip(170 + 25 * np.sin(xx / 5.0)[..., None] + 10 * rng.randn(size, size, 3), 0, 255)[sel]
    elif kind == "two":
        for cy, cx, r, v in ((0.3, 0.3, 0.16, 200), (0.68, 0.66, 0.2, 120)):
            sel = (yy - size * cy) ** 2 + (xx - size * cx) ** 2 < (size * r) ** 2
            img[sel] = np.clip(v + 12 * rng.randn(size, size, 3), 0, 255)[sel]
    elif kind == "ring":
        d = np.sqrt((yy - size / 2) ** 2 + (xx - size / 2) ** 2)
        sel = (d < size * 0.4) & (d > size * 0.22)
        img[sel] = np.clip(210 + 10 * rng.randn(size, size, 3), 0, 255)[sel]
    elif kind == "flat":
        img[:] = 77
    elif kind == "noise":
        img = rng.randint(0, 256, (size, size, 3)).astype(np.float64)
    return img.astype(np.uint8)


def test_obj_mask_kernels_match_the_host_statement(golden):
    """ops.obj_mask_batch (csrc/objmask.hip) against dataset_generator.obj_mask / _canny -- the host statement that is pinned to
    scikit-image 0.18.3 -- and against the scikit-image fixture itself: edge maps and object masks identical, bit for bit."""
    from self_supervised import dataset_generator as dg, ops
    dev = torch.device("cuda:0")
    d = golden("skimage")
    for k in range(int(d["n"])):
        img = d[f"img{k}"]
        mask, edges = ops.obj_mask_batch(torch.from_numpy(img[None]).to(dev), return_edges=True)
        assert np.array_equal(edges[0].cpu().numpy(), d[f"canny{k}"]), k
        assert np.array_equal(mask[0].cpu().numpy(), d[f"mask{k}"]), k
    rng = np.random.RandomState(5)
    for size in (64, 96, 256):
        imgs = np.stack([_blob_image(rng, size, kind) for kind in ("disc", "two", "ring", "flat", "noise", "disc")])
        mask, edges = ops.obj_mask_batch(torch.from_numpy(imgs).to(dev), return_edges=True, chunk=4)
        for i, im in enumerate(imgs):
            gray = np.array(Image.fromarray(im).convert("L"))
            want_e = dg._canny(gray, sigma=1.5, low=5, high=15)
            want_m = np.asarray(dg.obj_mask(Image.fromarray(im)).convert("1"))
            assert np.array_equal(edges[i].cpu().numpy(), want_e), (size, i)
            assert np.array_equal(mask[i].cpu().numpy(), want_m), (size, i)
    assert bool(mask[3].all())                       # a flat image has no edges: the reference's all-white mask


def test_screw_loader_builds_its_masks_on_the_device(tmp_path):
    """A non-fixed-object category: the GPU-resident loader's per-image object masks (device kernels) equal obj_mask on the host."""
    from fake_mvtec import make_tree
    from self_supervised import augment, datasets, dataset_generator as dg
    root = make_tree(str(tmp_path / "dataset"), categories=("screw",), n_train=6, n_test_good=1, n_test_bad=1, size=160)
    tr = sorted(os.path.join(root, "screw", "train/good", f) for f in os.listdir(os.path.join(root, "screw", "train/good")))
    ds = datasets.PretextTaskDataset("screw", np.array(tr), imsize=(128, 128), transform=datasets._default_transform(), dataset_root=root)
    ld = augment.GpuPretextLoader(ds, 4, num_workers=0)
    want = np.stack([np.asarray(dg.obj_mask(Image.open(n).resize((128, 128)).convert("RGB")).convert("1")) for n in tr])
    assert np.array_equal(ld.aug.masks, want)
