"""GPU-resident synthetic-defect pipeline: host-side parameter sampling + one HIP launch sequence per batch.

``sample_defect`` restates the sampling of ``PretextTaskDataset.__getitem__`` (src/self_supervised/datasets.py:
209-394 of the reference): label, optional affine / crop, defect source (crop 70 % / average colour 15 % / random
colour 15 %), polygon vertices, container clamp, scar angle and copies, poly-line points, jitter -- using the same
helpers (dataset_generator) and the same distributions.  It only produces numbers; ``GpuCutPaste.__call__`` sends
one record per sample to csrc/augment.hip, which does every per-pixel operation for the whole batch.
"""
import ctypes
import os
import random

import numpy as np
import torch
from scipy.signal import savgol_filter

from . import _hip, constants
from .dataset_generator import (check_valid_coordinates_by_container, get_random_coordinate, polygon_points,
                                sample_patch_box)
from .datasets import CPP, IMAGENET_MEAN, IMAGENET_STD
from . import pil_exact as px
from .tv_transforms import RandomAffine, RandomCrop, inverse_affine_matrix

MAX_LINE_POINTS = 32
AUG_DTYPE = np.dtype([
    ("label", "<i4"), ("crop_left", "<i4"), ("crop_top", "<i4"), ("aff_on", "<i4"), ("aff_fix", "<i4", 6),
    ("cut_index", "<i4"), ("cut_left", "<i4"), ("cut_top", "<i4"), ("cut_w", "<i4"), ("cut_h", "<i4"),
    ("patch_src_left", "<i4"), ("patch_src_top", "<i4"), ("patch_w", "<i4"), ("patch_h", "<i4"),
    ("patch_dst_left", "<i4"), ("patch_dst_top", "<i4"), ("patch_flat", "<i4"),
    ("patch_rgb", "<i4", 3), ("patch_bright", "<f4", 2), ("patch_nbright", "<i4"), ("poly_n", "<i4"), ("poly_xy", "<i4", 16),
    ("scar_src_left", "<i4"), ("scar_src_top", "<i4"), ("scar_w", "<i4"), ("scar_h", "<i4"), ("scar_flat", "<i4"),
    ("scar_rgb", "<i4", 3), ("scar_bright", "<f4", 2), ("scar_nbright", "<i4"), ("scar_rot", "<i4"), ("scar_fix", "<i4", 6),
    ("scar_rw", "<i4"), ("scar_rh", "<i4"), ("scar_n", "<i4"), ("scar_dst", "<i4", 10),
    ("line_n", "<i4"), ("line_xy", "<i4", 2 * MAX_LINE_POINTS), ("line_quad", "<i4", 8 * (MAX_LINE_POINTS - 1)),
    ("line_quad_ok", "<i4", MAX_LINE_POINTS - 1), ("line_rgb", "<i4", 3), ("line_width", "<i4"),
    ("jit_n", "<i4"), ("jit_order", "<i4", 3), ("jit_factor", "<f4", 3)])

_NAMED = {"black": (0, 0, 0), "white": (255, 255, 255), "silver": (192, 192, 192), "gray": (128, 128, 128)}


def _crop_mean_rgb(img_u8, box):
    """np.array(img.crop(box)).mean(axis=(0, 1)) -- Image.crop pads with zeros where the box leaves the image."""
    l, t, w, h = box
    H, W = img_u8.shape[:2]
    inside = img_u8[max(t, 0):min(t + h, H), max(l, 0):min(l + w, W)].reshape(-1, 3).astype(np.float64)
    return inside.sum(axis=0) / float(w * h)


def _cos(a, b):
    """dataset_generator.check_color_similarity on two mean colours."""
    a, b = np.asarray(a, float)[:3] / 255.0, np.asarray(b, float)[:3] / 255.0
    return float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b)))


def _source(rec, prefix, cut_u8, area_ratio, aspect_ratio):
    """crop / average colour / random colour choice + box, as generate_patch draws them.  Returns (w, h, mean colour of
    the patch image) -- the mean check_color_similarity sees."""
    t = np.random.choice([0, 1, 2], p=[0.7, 0.15, 0.15])
    l, tp, w, h = sample_patch_box((cut_u8.shape[1], cut_u8.shape[0]), area_ratio, aspect_ratio)
    rec[prefix + "_src_left"], rec[prefix + "_src_top"], rec[prefix + "_w"], rec[prefix + "_h"] = l, tp, w, h
    if t == 2:
        rgb = (random.randint(0, 255), random.randint(0, 255), random.randint(0, 255))
    elif t == 1:
        mean = _crop_mean_rgb(cut_u8, (l, tp, w, h))
        rgb = (int(mean[0]), int(mean[1]), int(mean[2]))
    else:
        rgb = None
    rec[prefix + "_flat"] = int(rgb is not None)
    if rgb is not None:
        rec[prefix + "_rgb"] = rgb
        return w, h, np.asarray(rgb, float)
    return w, h, _crop_mean_rgb(cut_u8, (l, tp, w, h))


def _decorrelate(rec, prefix, x_mean, src_mean):
    rec[prefix + "_bright"], rec[prefix + "_nbright"] = (1.0, 1.0), 0
    if _cos(x_mean, src_mean) > 0.99:
        low, high = np.random.uniform(0.75, 0.9), np.random.uniform(1.1, 1.15)
        rec[prefix + "_bright"], rec[prefix + "_nbright"] = (random.choice([low, high]), random.choice([low, high])), 2


def _window_sum(img_u8, fix, left, top, w, h):
    """Channel sums (exact integers) of a window of the NEAREST-affine-transformed image: the library's host helper (one C loop
    over the window) instead of warping the whole image in numpy -- 4.5 ms -> 0.1 ms per call at 256 x 256, and it was 75 % of
    the sampler's time."""
    H, W = img_u8.shape[:2]
    img = np.ascontiguousarray(img_u8)
    out = (ctypes.c_int64 * 3)()
    cf = None if fix is None else (ctypes.c_int32 * 6)(*[int(v) for v in fix])
    _hip.check(_hip.lib().ssad_affine_window_sum_u8(img.ctypes.data, H, W, cf, int(left), int(top), int(w), int(h), out))
    return np.array([out[0], out[1], out[2]], dtype=np.float64)


def mask_coordinates(seg):
    """(x, y) of the mask's pixels in row-major order: datasets.py:283's coords_map."""
    return np.flip(np.column_stack(np.where(seg)), axis=1)


def sample_defect(subject, img_u8, seg_mask, cuts_u8=None, patch_localization=False, patch_size=64, coords=None):
    """One ssad_aug_params record (numpy void) for an H x W uint8 image and its boolean object mask: every random draw of
    PretextTaskDataset.__getitem__ in its order (pinned by tests/test_data_cpu.py), every geometric quantity as Pillow
    computes it (pil_exact).  coords: mask_coordinates(seg_mask) when the caller keeps it (image level only)."""
    H, W = img_u8.shape[:2]
    rec = np.zeros((), AUG_DTYPE)
    rec["cut_index"] = -1
    y = random.randint(0, 3)
    aff = None
    if not patch_localization and subject not in constants.NON_FIXED_OBJECTS():
        ang, sc = RandomAffine(3, scale=(1.05, 1.1)).sample()
        aff = inverse_affine_matrix((W * 0.5, H * 0.5), ang, (0, 0), sc)
        assert not px.affine_is_scale_only(aff), "a rotation of exactly 0 degrees takes another Pillow code path"
        rec["aff_on"], rec["aff_fix"] = 1, px.affine_fix_coeffs(aff)
    cut_u8 = img_u8
    if subject in constants.TEXTURES() and cuts_u8 is not None and len(cuts_u8):
        ci = random.randrange(len(cuts_u8))
        rec["cut_index"], cut_u8 = ci, cuts_u8[ci]
    h, w, k_patch, k_scar = H, W, 1.75, 2
    rec["cut_w"], rec["cut_h"] = W, H
    seg = seg_mask
    if patch_localization:
        ps = patch_size
        # datasets.py:244-249: capsule and screw are first cut to a fixed window (Image.crop pads with zeros where the window
        # leaves the image -- the kernel's out-of-image fetches and the mask crop below do the same)
        ox, oy, ww, wh = {"capsule": (0, 50, 255, 150), "screw": (25, 25, 205, 205)}.get(subject, (0, 0, W, H))
        left, top = random.randint(0, ww - ps), random.randint(0, wh - ps)
        left, top = ox + left, oy + top
        rec["crop_left"], rec["crop_top"] = left, top
        seg = np.zeros((ps, ps), bool)
        ys, xs = slice(top, min(top + ps, H)), slice(left, min(left + ps, W))
        if top < H and left < W:
            seg[:ys.stop - ys.start, :xs.stop - xs.start] = seg_mask[ys, xs]
        ct, cl = RandomCrop(ps).sample(W, H)
        rec["cut_left"], rec["cut_top"], rec["cut_w"], rec["cut_h"] = cl, ct, ps, ps
        cut_u8 = cut_u8[ct:ct + ps, cl:cl + ps]
        h = w = ps
        k_patch = k_scar = 1
        # the reference sums ToTensor() of the RGB mask crop (datasets.py:258): a white pixel counts three times
        if 3 * int(seg.sum()) < int((ps * ps) / 2):
            y = 0
    area_p = CPP.rectangle_area_ratio_patch if patch_localization else CPP.rectangle_area_ratio
    area_s = CPP.scar_area_ratio_patch if patch_localization else CPP.scar_area_ratio
    if y > 0:
        coords_map = coords if (coords is not None and seg is seg_mask) else mask_coordinates(seg)

    def x_mean():
        """Mean colour of the image the defect is pasted into (after RandomAffine / crop), for the similarity test; the zero
        padding outside the image counts in the mean."""
        fix = rec["aff_fix"] if aff is not None else None
        return _window_sum(img_u8, fix, int(rec["crop_left"]), int(rec["crop_top"]), w, h) / float(h * w)
    if y == 1:
        centre = get_random_coordinate(coords_map)
        pw, ph, mean = _source(rec, "patch", cut_u8, area_p, CPP.rectangle_aspect_ratio)
        _decorrelate(rec, "patch", x_mean(), mean)
        at = check_valid_coordinates_by_container((w, h), (pw, ph), current_coords=centre, container_scaling_factor=k_patch)
        rec["patch_dst_left"], rec["patch_dst_top"] = at
        pts = polygon_points((pw, ph), sides=8)
        rec["poly_n"] = len(pts)
        rec["poly_xy"][:2 * len(pts)] = np.asarray(pts, np.int32).ravel()
    elif y == 2:
        sw, sh, mean = _source(rec, "scar", cut_u8, area_s, CPP.scar_aspect_ratio)
        _decorrelate(rec, "scar", x_mean(), mean)
        copies, angle = random.randint(2, 5), random.randint(-45, 45)
        rw, rh, m = px.rotate_params(sw, sh, angle)
        rec["scar_rw"], rec["scar_rh"], rec["scar_n"] = rw, rh, copies
        if m is not None:
            assert not px.affine_is_scale_only(m)
            rec["scar_rot"], rec["scar_fix"] = 1, px.affine_fix_coeffs(m)
        for k in range(copies):
            centre = get_random_coordinate(coords_map)
            at = check_valid_coordinates_by_container((w, h), (rw, rh), current_coords=centre, container_scaling_factor=k_scar)
            rec["scar_dst"][2 * k:2 * k + 2] = at
    elif y == 3:
        side = random.choice(['left', 'top'])
        n = 30 if patch_localization else 60
        pts, c = [], 0
        for i in range(n):
            idx = random.randint(c, int(len(coords_map) * (i / n)))
            pts.append(tuple(coords_map[idx]))
            c = idx
        rgb = _NAMED[random.choice(['black', 'white', 'silver'])]
        if side == 'left':
            pts.sort(key=lambda t: t[0])
        pts = savgol_filter(pts, 10, 2, axis=0)
        if not patch_localization:
            pts = np.array_split(pts, 10)[random.randint(0, 9)]
        width = 1 if patch_localization else 3
        ip = px.line_points_int([tuple(p) for p in pts])
        assert len(ip) <= MAX_LINE_POINTS, "poly-line longer than the kernel's record"
        rec["line_n"] = len(ip)
        rec["line_xy"][:2 * len(ip)] = np.asarray(ip, np.int32).ravel()
        rec["line_rgb"], rec["line_width"] = rgb, width
        if width > 1:
            for k, ((x0, y0), (x1, y1)) in enumerate(zip(ip[:-1], ip[1:])):
                q = px.wide_line_quad(x0, y0, x1, y1, width)
                if q is not None:
                    rec["line_quad_ok"][k] = 1
                    rec["line_quad"][8 * k:8 * k + 8] = np.asarray(q, np.int32).ravel()
    rec["label"] = y
    order, f = CPP.jitter_transforms.sample()
    rec["jit_n"] = len(order)
    rec["jit_order"][:len(order)] = order
    rec["jit_factor"] = f
    return rec, (h, w)


class DefectSampler:
    """Host half of the pipeline: one ssad_aug_params record per sample from the category's uint8 images and object masks.
    Holds numpy arrays only (no device, no torch state), so a sampler worker rebuilds it from arrays mapped out of shared memory."""

    def __init__(self, subject, images_u8, seg_masks, cuts_u8=None, patch_localization=False, patch_size=64, mask_index=None):
        self.subject, self.patch_localization, self.patch_size = subject, patch_localization, patch_size
        self.images_cpu = images_u8 if isinstance(images_u8, np.memmap) else np.ascontiguousarray(images_u8, dtype=np.uint8)
        self.cuts_cpu = None if cuts_u8 is None else (cuts_u8 if isinstance(cuts_u8, np.memmap) else
                                                        np.ascontiguousarray(cuts_u8, dtype=np.uint8))
        masks = np.asarray(seg_masks, dtype=bool)
        if masks.ndim == 2:
            masks = masks[None]
        n = self.images_cpu.shape[0]
        if mask_index is None:
            # one mask for the whole category (fixed objects, textures) or one per image (screw, ...): identical masks are kept once
            if masks.shape[0] == 1 or (masks.strides[0] == 0):
                masks, mask_index = np.ascontiguousarray(masks[:1]), np.zeros(n, np.int64)
            else:
                assert masks.shape[0] == n
                uniq, mask_index, seen = [], np.zeros(n, np.int64), {}
                for i in range(n):
                    k = masks[i].tobytes()
                    if k not in seen:
                        seen[k] = len(uniq)
                        uniq.append(masks[i])
                    mask_index[i] = seen[k]
                masks = np.stack(uniq)
        self.masks_unique, self.mask_index = masks, np.asarray(mask_index, dtype=np.int64)
        self._coords = {}

    def mask_of(self, i):
        return self.masks_unique[self.mask_index[i]]

    @property
    def masks(self):
        return self.masks_unique[self.mask_index]

    def sample(self, indices):
        """One record per index, drawn from the process-global python / numpy / torch RNG streams in the order
        PretextTaskDataset.__getitem__ draws.  -> (records [B] of AUG_DTYPE, (h, w) of the synthesised images)."""
        recs, hw = [], None
        for i in np.asarray(indices, dtype=np.int64):
            r, hw = sample_defect(self.subject, self.images_cpu[i], self.mask_of(i), self.cuts_cpu, self.patch_localization,
                                  self.patch_size, coords=self._coords_of(i))
            recs.append(r)
        return np.stack(recs), hw

    def _coords_of(self, i):
        """mask_coordinates of image i's object mask, computed once per distinct mask (fixed-object categories share one; for
        per-image masks the cache keeps the most recent 64)."""
        if self.patch_localization:
            return None                      # the mask is cropped per sample there
        key = int(self.mask_index[i])
        c = self._coords.get(key)
        if c is None:
            if len(self._coords) >= 64:
                self._coords.pop(next(iter(self._coords)))
            c = self._coords[key] = mask_coordinates(self.masks_unique[key])
        return c

    # ---- publication to sampler workers: arrays as .npy files in shared memory, mapped read-only by the workers ----
    def publish(self):
        """-> (descriptor, paths): a small picklable description a worker rebuilds this sampler from."""
        import tempfile
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        d = tempfile.mkdtemp(prefix="ssad_sampler_", dir=base)
        files = {}
        for name, arr in (("images", self.images_cpu), ("masks", self.masks_unique), ("mask_index", self.mask_index),
                          ("cuts", self.cuts_cpu)):
            if arr is not None:
                files[name] = os.path.join(d, name + ".npy")
                np.save(files[name], np.ascontiguousarray(arr))
        desc = {"key": d, "subject": self.subject, "patch_localization": self.patch_localization, "patch_size": self.patch_size,
                "files": files}
        if not _PUBLISHED:
            import atexit
            atexit.register(_remove_published)          # a loader that is never closed must not leave its arrays in /dev/shm
        _PUBLISHED.add(d)
        return desc, d

    @staticmethod
    def attach(desc):
        f = desc["files"]
        ld = lambda k: np.load(f[k], mmap_mode="r") if k in f else None
        return DefectSampler(desc["subject"], ld("images"), np.asarray(ld("masks")), ld("cuts"), desc["patch_localization"],
                             desc["patch_size"], mask_index=np.asarray(ld("mask_index")))


_PUBLISHED = set()      # shared-memory directories of this process's published samplers


def _remove_published():
    import shutil
    for d in list(_PUBLISHED):
        shutil.rmtree(d, ignore_errors=True)
        _PUBLISHED.discard(d)


class GpuCutPaste:
    """Batch augmenter: images stay on the GPU as uint8 HWC; returns (x fp32 NCHW normalised, y int64, original fp32)."""

    def __init__(self, subject, images_u8, seg_masks, cuts_u8=None, patch_localization=False, patch_size=64, device="cuda"):
        assert _hip.lib().ssad_aug_params_size() == AUG_DTYPE.itemsize, "ssad_aug_params layout drifted from augment.py"
        self.subject, self.patch_localization, self.patch_size = subject, patch_localization, patch_size
        self.host = DefectSampler(subject, images_u8, seg_masks, cuts_u8, patch_localization, patch_size)
        self.images_cpu, self.cuts_cpu = self.host.images_cpu, self.host.cuts_cpu
        self.device = torch.device(device)
        self.images = torch.from_numpy(self.images_cpu).to(self.device)
        self.cuts = None if self.cuts_cpu is None else torch.from_numpy(self.cuts_cpu).to(self.device)
        self._mean = (ctypes.c_float * 3)(*IMAGENET_MEAN)
        self._std = (ctypes.c_float * 3)(*IMAGENET_STD)

    @property
    def masks(self):
        return self.host.masks

    def sample(self, indices):
        """Host half (DefectSampler.sample) in this process."""
        return self.host.sample(indices)

    def __call__(self, indices):
        idx = np.asarray(indices, dtype=np.int64)
        recs, hw = self.sample(idx)
        return self.synthesise(idx, recs, hw)

    def synthesise(self, idx, recs, hw):
        """Device half: every per-pixel operation of the batch (csrc/augment.hip) from the records."""
        b, (h, w) = len(idx), hw
        _, H, W, _ = self.images.shape
        params = torch.from_numpy(np.array(recs.view(np.uint8).reshape(b, -1))).to(self.device)   # records from a worker are read-only
        batch = self.images[torch.from_numpy(idx).to(self.device)].contiguous()
        work = torch.empty((b, h, w, 3), dtype=torch.uint8, device=self.device)
        gmean = torch.empty(b, dtype=torch.float32, device=self.device)
        out = torch.empty((b, 3, h, w), dtype=torch.float32, device=self.device)
        orig = torch.empty((b, 3, H, W), dtype=torch.float32, device=self.device)
        lib = _hip.lib()
        _hip.check(lib.ssad_cutpaste_augment(batch.data_ptr(), None if self.cuts is None else self.cuts.data_ptr(),
                                             params.data_ptr(), work.data_ptr(), gmean.data_ptr(), out.data_ptr(), b, H, W, h, w,
                                             self._mean, self._std, _hip.stream()))
        _hip.check(lib.ssad_u8hwc_to_f32chw(batch.data_ptr(), orig.data_ptr(), b, H, W, _hip.stream()))
        y = torch.from_numpy(recs["label"].astype(np.int64)).to(self.device)
        return out, y, orig


# ---------------------------------------------------------------------------------------------
# sampler workers
# ---------------------------------------------------------------------------------------------
_POOL_STATE = {}      # key -> sampler (DefectSampler / GpuCutPaste) of THIS process: in-process lookups and, in a worker, the
                      # samplers it has attached so far
_POOL = {"exe": None, "size": 0, "ctx": None}


def _batch_seed(base, epoch, batch_no, rank, stage=0):
    """Seed of one batch: a function of (loader base seed, stage, epoch, batch number, rank) only.  `stage` counts how often the
    loader's epoch counter has restarted (tools.training fits twice -- projection head, then fine tuning -- and each fit numbers its
    epochs from 0): without it the second fit would replay the first one's synthetic batches."""
    s = (int(base) * 1000003 + int(stage)) if stage else int(base)
    return (((s * 1000003 + int(epoch)) * 1000003 + int(batch_no)) * 1000003 + int(rank)) % (2 ** 63)


def _sampler_for(desc):
    if not isinstance(desc, dict):
        return _POOL_STATE[desc]
    smp = _POOL_STATE.get(desc["key"])
    if smp is None:
        if len(_POOL_STATE) >= 8:            # a worker outlives the loaders it served: drop the oldest mappings
            _POOL_STATE.pop(next(iter(_POOL_STATE)))
        smp = _POOL_STATE[desc["key"]] = DefectSampler.attach(desc)
    return smp


def _pool_sample(desc, seed, idx):
    """Runs in a sampler worker: the records of one batch under that batch's own seed.  desc: a key of _POOL_STATE or the
    descriptor DefectSampler.publish returned."""
    smp = _sampler_for(desc)
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    recs, hw = smp.sample(idx)
    return recs.tobytes(), hw


def _pool_warm(_):
    return os.getpid()


def sampler_pool(num_workers):
    """The process-wide pool of sampler workers (created on first use, grown when a loader asks for more workers, shut down at
    exit).  Workers hold no GPU state: they are started through datasets._worker_context() -- plain fork while this process has not
    touched the GPU, a fork SERVER afterwards (fork()ing a process with live HIP queues stalls them: datasets._worker_context) --
    and receive a loader's arrays through shared-memory files, not through the fork."""
    from concurrent.futures import ProcessPoolExecutor
    from .datasets import _hidden_main, _worker_context
    if _POOL["exe"] is not None and _POOL["size"] >= num_workers:
        return _POOL["exe"]
    if _POOL["exe"] is not None:
        # grow: the old pool finishes what it was given (other live loaders may still hold its futures -- their speculative next-epoch
        # batches -- and a cancelled future would surface as CancelledError in the middle of their epoch); its workers exit afterwards
        _POOL["exe"].shutdown(wait=False)
    ctx = _worker_context(preload=("self_supervised.augment",))
    exe = ProcessPoolExecutor(num_workers, mp_context=ctx)
    with _hidden_main(ctx):
        list(exe.map(_pool_warm, range(4 * num_workers)))       # start the workers now (imports done before the first epoch)
    if _POOL["exe"] is None:
        import atexit
        atexit.register(_shutdown_pool)
    _POOL.update(exe=exe, size=num_workers, ctx=ctx)
    return exe


def _shutdown_pool():
    if _POOL["exe"] is not None:
        _POOL["exe"].shutdown(wait=False, cancel_futures=True)
        _POOL.update(exe=None, size=0)


class GpuPretextLoader:
    """GPU-resident replacement for ``DataLoader(PretextTaskDataset)``: the category's images are decoded and resized
    once, kept on the GPU as uint8, and every batch is synthesised there (GpuCutPaste).  Yields the same triple the
    Dataset does -- (x normalised fp32, y int64, original fp32 in [0,1]) -- already on the device.

    Host side of a batch = drawing one parameter record per sample (0.4 ms each: label, boxes, polygon, scar, poly-line, jitter
    -- numbers only).  ``num_workers = 0``: drawn in this process from the global RNG streams, sample after sample in the
    order the reference's ``__getitem__`` draws (what the parity tests pin).  ``num_workers > 0`` (the counterpart of the
    reference's ``DataLoader(num_workers=8)``, src/self_supervised/datasets.py:501-533): sampler processes draw whole
    batches ahead of the training loop, batch b of epoch e under its own seed (base_seed, stage, e, b, rank), so the stream does not
    depend on the number of workers or on their timing; the device work of a batch is unchanged.  Measured: 2.4 k samples/s per
    worker at 256 x 256 image level; 8 workers keep up with the 7.4 k img/s of the training step.

    Epochs: ``shard(world, rank, epoch)`` / ``set_epoch(epoch)`` name the epoch about to be iterated; when the counter does not
    advance (a second ``fit`` numbering its epochs from 0 again) the loader's ``stage`` moves on, so no two passes over the loader
    ever share a shuffle or a batch seed -- the reference draws fresh augmentations continuously."""

    def __init__(self, dataset, batch_size, shuffle=True, drop_last=True, device="cuda", rank=0, world=1, num_workers=0,
                 base_seed=0, prefetch_batches=None):
        from PIL import Image
        from .dataset_generator import obj_mask
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.rank, self.world, self.epoch = rank, world, 0
        self.stage, self._last_epoch = 0, None
        names = list(dict.fromkeys(dataset.images_filenames))            # duplicated file lists decode once
        self.slot = {n: i for i, n in enumerate(names)}
        # decode on the host (threads), Pillow's bicubic resize on the device (gpu_io / csrc/resize.hip: bit-exact), one copy back
        # for the host-side sampler; a CPU `device` (tests of the host half) keeps Pillow's resize
        if torch.device(device).type == "cuda":
            from .gpu_io import load_rgb_batch
            imgs = list(load_rgb_batch(names, dataset.imsize, device).cpu().numpy())
        else:
            imgs = [np.asarray(Image.open(n).resize(dataset.imsize).convert('RGB')) for n in names]
        if dataset.subject in constants.NON_FIXED_OBJECTS():
            if torch.device(device).type == "cuda":          # Canny -> morphology -> largest component on the device (csrc/objmask.hip)
                from . import ops
                masks = ops.obj_mask_batch(torch.from_numpy(np.stack(imgs)).to(device)).cpu().numpy()
            else:
                masks = np.stack([np.asarray(obj_mask(Image.fromarray(im)).convert('1')) for im in imgs])
        else:
            masks = np.asarray(dataset.fixed_segmentation.convert('1'))
        cuts = np.stack([np.asarray(c) for c in dataset.images_for_cut]) if dataset.subject in constants.TEXTURES() else None
        self.aug = GpuCutPaste(dataset.subject, np.stack(imgs), masks, cuts, dataset.patch_localization,
                               dataset.patch_size, device)
        self.index = np.array([self.slot[n] for n in dataset.images_filenames])
        self.num_workers = int(num_workers)
        self.base_seed = int(base_seed)
        self.prefetch_batches = prefetch_batches if prefetch_batches is not None else 2 * max(self.num_workers, 1)
        self._desc, self._dir = None, None
        self._side = None
        self._spec, self.speculate, self.spec_hits = None, os.environ.get("SSAD_LOADER_SPECULATE", "1") != "0", 0

    def set_epoch(self, epoch):
        if self._last_epoch is not None and epoch <= self._last_epoch:
            self.stage += 1
        self.epoch = self._last_epoch = int(epoch)

    def shard(self, world, rank, epoch):
        self.world, self.rank = world, rank
        self.set_epoch(epoch)
        return self

    def __len__(self):
        n = len(self.index) // self.world
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _batches(self, epoch=None):
        epoch = self.epoch if epoch is None else epoch
        order = np.arange(len(self.index))
        if self.shuffle:
            order = np.random.RandomState((1234 + epoch + 7919 * self.stage) % (2 ** 32)).permutation(len(order))
        order = order[self.rank::self.world]
        return [self.index[order[i * self.batch_size:(i + 1) * self.batch_size]] for i in range(len(self))]

    def _published(self):
        if self._desc is None:
            self._desc, self._dir = self.aug.host.publish()
        return self._desc

    def close(self):
        """Removes the shared-memory copy of the category (workers that still map it keep their mapping until they drop it)."""
        spec, self._spec = getattr(self, "_spec", None), None
        if spec is not None:
            for _, fut in spec["pending"]:
                fut.cancel()
        if self._dir is not None:
            import shutil
            shutil.rmtree(self._dir, ignore_errors=True)
            _PUBLISHED.discard(self._dir)
            self._desc = self._dir = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __iter__(self):
        batches = self._batches()
        if self.num_workers <= 0:
            for idx in batches:
                yield self.aug(idx)
            return
        from .datasets import _hidden_main
        pool, desc = sampler_pool(self.num_workers), self._published()
        key = (self.stage, self.epoch, self.rank, self.world)
        # The sampling of a batch takes one worker ~45 ms (96 images): at the start of an epoch nothing is in flight and the training
        # stream would sit idle for that long, every epoch.  So once every batch of THIS epoch has been handed to the pool, the
        # hand-out goes on with the first batches of the NEXT epoch of the same stage (its order and seeds are functions of the epoch
        # number); the next __iter__ takes them over when it is that epoch, and cancels them when it is not (a new fit, the end).
        spec, self._spec = getattr(self, "_spec", None), None
        if spec is not None and spec["key"] == key:
            pending, nxt = spec["pending"], spec["nxt"]
            self.spec_hits += 1
        else:
            if spec is not None:
                for _, fut in spec["pending"]:
                    fut.cancel()
            pending, nxt = [], 0
        nxt_batches = nxt_pending = None
        nxt_n = 0

        def submit():
            nonlocal nxt, nxt_batches, nxt_pending, nxt_n
            if nxt < len(batches):
                with _hidden_main(_POOL["ctx"]):
                    fut = pool.submit(_pool_sample, desc, _batch_seed(self.base_seed, self.epoch, nxt, self.rank, self.stage),
                                      batches[nxt])
                pending.append((batches[nxt], fut))
                nxt += 1
                return
            if not self.speculate:
                return
            if nxt_batches is None:
                nxt_batches, nxt_pending = self._batches(self.epoch + 1), []
                self._spec = {"key": (self.stage, self.epoch + 1, self.rank, self.world), "pending": nxt_pending, "nxt": 0}
            if nxt_n < len(nxt_batches) and nxt_n < max(1, self.num_workers):
                with _hidden_main(_POOL["ctx"]):
                    fut = pool.submit(_pool_sample, desc, _batch_seed(self.base_seed, self.epoch + 1, nxt_n, self.rank, self.stage),
                                      nxt_batches[nxt_n])
                nxt_pending.append((nxt_batches[nxt_n], fut))
                nxt_n += 1
                self._spec["nxt"] = nxt_n
        for _ in range(self.prefetch_batches - len(pending)):
            submit()
        # Device half one batch ahead, on a side stream: the uploads of a batch's records / indices / labels come from pageable host
        # memory and synchronise the stream they are issued on -- on the training stream the host would wait for the step in flight
        # before it could even begin to prepare the next batch (measured: 1.9 ms of a 16.6 ms step at batch 96).  On the side stream
        # they wait for nothing, the synthesis kernels run beside the training step, and the consumer's stream is made to wait for
        # the batch's event before it reads the tensors.
        side = self._side if getattr(self, "_side", None) is not None else torch.cuda.Stream(self.aug.device)
        self._side = side

        def produce():
            idx, fut = pending.pop(0)
            raw, hw = fut.result()
            submit()
            with torch.cuda.stream(side):
                out = self.aug.synthesise(np.asarray(idx, dtype=np.int64), np.frombuffer(raw, dtype=AUG_DTYPE), hw)
                ev = torch.cuda.Event()
                ev.record(side)
            return out, ev
        try:
            ahead = produce() if pending else None
            while ahead is not None:
                out, ev = ahead
                ahead = produce() if pending else None
                cur = torch.cuda.current_stream(self.aug.device)
                cur.wait_event(ev)
                for t in out:
                    t.record_stream(cur)
                yield out
        finally:
            for _, fut in pending:           # an epoch the consumer left early (limit_*_batches): what has not started is dropped
                fut.cancel()
