"""Datasets / datamodules with the reference's surface (src/self_supervised/datasets.py:33-533).

``PretextTaskDataset[i] -> (x (3,h,w) f32 ImageNet-normalised, y in {0..3}, original (3,H,W) f32 in [0,1])`` and
``MVTecDataset[i] -> (x, gt (1,H,W), original)``.  torchvision / pytorch_lightning are not required: the few
transforms the reference takes from torchvision (ToTensor, Normalize, ColorJitter, RandomAffine, RandomCrop) are
restated on PIL below, and the datamodules are plain classes exposing the same ``*_dataloader()`` methods.

The defect synthesis itself has two back-ends sharing one parameter sampler (``sample_defect``):
  * PIL (this file; per-sample, CPU, what ``__getitem__`` returns -- the reference's path);
  * HIP (augment.py / csrc/augment.hip; whole batches resident on the GPU).
Dataset root is injectable (``dataset_root``); the reference hard-codes ``'dataset/'`` (datasets.py:189-200).
"""
import os
import random

import numpy as np
import torch
from PIL import Image, ImageDraw, ImageEnhance
from scipy.signal import savgol_filter
from torch.utils.data import DataLoader, Dataset

from . import constants
from .dataset_generator import (check_color_similarity, check_valid_coordinates_by_container, generate_patch,
                                get_random_coordinate, label_mean_rgb, obj_mask, paste_patch, rect2poly, slic_superpixels,
                                slic_superpixels_cached)
from .functional import (duplicate_filenames, get_all_subject_experiments, get_filenames, get_ground_truth,
                         get_ground_truth_filename, get_test_data_filenames)

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


# transforms: third-party restatements (torchvision.transforms on PIL images) live in tv_transforms.py
from .tv_transforms import ColorJitter, Compose, Normalize, RandomAffine, RandomCrop, ToTensor, to_tensor  # noqa: E402,F401


class CPP:
    """Cut-paste constants (datasets.py:33-47)."""
    jitter_offset = 0.1
    rectangle_area_ratio_patch = (0.2, 0.5)
    rectangle_area_ratio = (0.03, 0.07)
    rectangle_aspect_ratio = ((0.3, 0.5), (1, 3.3))
    scar_area_ratio_patch = (0.02, 0.05)
    scar_area_ratio = (0.003, 0.007)
    scar_aspect_ratio = ((0.3, 0.5), (2.5, 3.3))
    jitter_transforms = ColorJitter(brightness=jitter_offset, contrast=jitter_offset, saturation=jitter_offset)


def _default_transform():
    return Compose([ToTensor(), Normalize(IMAGENET_MEAN, IMAGENET_STD)])


# ---------------------------------------------------------------------------------------------
# MVTec test data (datasets.py:50-163)
# ---------------------------------------------------------------------------------------------
class MVTecDataset(Dataset):
    def __init__(self, dataset_dir: str, images_filenames: list, imsize: tuple = (256, 256), transform=None,
                 patch_level: bool = False) -> None:
        super().__init__()
        self.dataset_dir, self.images_filenames, self.imsize = dataset_dir, images_filenames, imsize
        self.transform, self.patch_level = transform, patch_level

    def __getitem__(self, index):
        filename = self.images_filenames[index]
        original = Image.open(filename).resize(self.imsize).convert('RGB')
        gt = get_ground_truth(get_ground_truth_filename(filename, self.dataset_dir + 'ground_truth/'), self.imsize)
        x = self.transform(original) if self.transform else original.copy()
        return x, to_tensor(gt), to_tensor(original)

    def __len__(self):
        return len(self.images_filenames)


def _worker_context(preload=()):
    """How DataLoader workers (and the sampler workers of augment.GpuPretextLoader) are started.  fork()ing a process that has initialised the GPU is what the reference's
    DataLoader(num_workers=8) does on Linux, and on this ROCm stack every such fork leaves the parent's GPU queues stalling for
    ~0.1 s at a time afterwards (copy-on-write faults on pages the driver has registered; `tools.inference` measured 10-13 s per
    call and growing with 2 workers against 1.6 s with none, round 3).  Once the GPU is up, workers therefore come from a fork
    SERVER (a clean helper process started once, torch preloaded); SSAD_LOADER_CONTEXT=fork|forkserver|spawn overrides.
    (_Loader below keeps unguarded driver scripts working with such workers.)"""
    import multiprocessing as mp
    name = os.environ.get("SSAD_LOADER_CONTEXT")
    if not name:
        name = "forkserver" if torch.cuda.is_initialized() else "fork"
    if name == "forkserver":
        try:
            mp.set_forkserver_preload(["torch", "numpy", "PIL.Image", "self_supervised.datasets", *preload])
        except Exception:
            pass
    return mp.get_context(name)


class _hidden_main:
    """While fork-server / spawn workers are being started, hide the caller's ``__main__`` from their bootstrap: the reference's
    driver scripts (src/test_training.py, ...) call ``training(...)`` at module level without an ``if __name__ == "__main__"``
    guard, which is fine under fork() and would re-run the whole script inside every worker otherwise.  What the workers unpickle
    lives in this package, so ``__main__`` is not needed there."""

    def __init__(self, ctx):
        self.active = ctx is not None and ctx.get_start_method() != "fork"

    def __enter__(self):
        import sys
        self.main = sys.modules.get("__main__") if self.active else None
        self.saved = {}
        if self.main is not None:
            self.saved = {k: self.main.__dict__[k] for k in ("__spec__", "__file__") if k in self.main.__dict__}
            self.main.__dict__["__spec__"] = None
            self.main.__dict__.pop("__file__", None)
        return self

    def __exit__(self, *exc):
        if self.main is not None:
            self.main.__dict__.update(self.saved)
        return False


class _Loader(DataLoader):
    """DataLoader whose fork-server / spawn workers do not re-import the caller's ``__main__`` (see _hidden_main)."""

    def __iter__(self):
        if not self.num_workers:
            return super().__iter__()
        with _hidden_main(self.multiprocessing_context):
            return super().__iter__()


class _DataModule:
    num_workers = 8

    def _loader(self, ds, shuffle, drop_last=False):
        nw = min(self.num_workers, os.cpu_count() or 1)
        return _Loader(ds, batch_size=self.batch_size, shuffle=shuffle, drop_last=drop_last, num_workers=nw,
                       persistent_workers=False, multiprocessing_context=_worker_context() if nw else None)

    def prepare_data(self) -> None:
        pass


class MVTecDatamodule(_DataModule):
    def __init__(self, root_dir: str, imsize: tuple = (256, 256), batch_size: int = 32, seed: int = 0, patch_level=None):
        self.root_dir, self.imsize, self.batch_size, self.seed, self.patch_level = root_dir, imsize, batch_size, seed, patch_level
        self.transform = _default_transform()
        self.train_images_filenames = get_filenames(self.root_dir + '/train/good/')
        self.test_images_filenames = get_test_data_filenames(self.root_dir + '/test/')

    def setup(self, stage=None) -> None:
        from sklearn.model_selection import train_test_split as tts
        tr, va = tts(self.train_images_filenames, test_size=0.2, random_state=self.seed)
        mk = lambda names: MVTecDataset(self.root_dir, names, imsize=self.imsize, transform=self.transform)
        self.train_dataset, self.val_dataset, self.test_dataset = mk(tr), mk(va), mk(self.test_images_filenames)

    def train_dataloader(self):
        return self._loader(self.train_dataset, True)

    def val_dataloader(self):
        return self._loader(self.val_dataset, False)

    def test_dataloader(self):
        return self._loader(self.test_dataset, False)

    def predict_dataloader(self):
        return self._loader(self.test_dataset, False)


# ---------------------------------------------------------------------------------------------
# pretext task (datasets.py:166-394)
# ---------------------------------------------------------------------------------------------
class PretextTaskDataset(Dataset):
    def __init__(self, subject: str, images_filenames, imsize: tuple = (256, 256), transform=None,
                 patch_localization: bool = False, patch_size: tuple = 64, dataset_root: str = 'dataset/') -> None:
        super().__init__()
        self.subject, self.images_filenames, self.imsize = subject, images_filenames, imsize
        self.transform, self.patch_localization, self.patch_size = transform, patch_localization, patch_size
        self.patch_area_ratio = CPP.rectangle_area_ratio_patch if patch_localization else CPP.rectangle_area_ratio
        self.scar_area_ratio = CPP.scar_area_ratio_patch if patch_localization else CPP.scar_area_ratio
        first = lambda sub: Image.open(dataset_root + sub + '/train/good/000.png').resize(self.imsize).convert('RGB')
        self.images_for_cut = [first(sub) for sub in get_all_subject_experiments(dataset_root)]
        if self.subject in constants.TEXTURES():
            self.fixed_segmentation = Image.new(size=self.imsize, mode='RGB', color='white')
        else:
            temp = first(self.subject)
            if self.subject == 'cable':
                # datasets.py:201-206: SLIC super-pixels (5 segments, sigma 2, Lab) painted with their mean colours first
                # (scikit-image is not a dependency here: dataset_generator.slic_superpixels restates it, pinned against the library)
                arr = np.array(temp)
                temp = Image.fromarray(label_mean_rgb(slic_superpixels_cached(arr, n_segments=5, sigma=2), arr)).convert('RGB')
            self.fixed_segmentation = obj_mask(temp)

    # -- one synthetic defect, PIL back-end --
    def _defect_source(self, cutting, area_ratio, aspect_ratio):
        t = np.random.choice([0, 1, 2], p=[0.7, 0.15, 0.15])
        kw = {} if t == 0 else {"colorized": True, "color_type": 'average' if t == 1 else 'random'}
        return generate_patch(cutting, area_ratio=area_ratio, aspect_ratio=aspect_ratio, **kw)

    @staticmethod
    def _decorrelate(x, patch):
        if check_color_similarity(x, patch) > 0.99:
            low, high = np.random.uniform(0.75, 0.9), np.random.uniform(1.1, 1.15)
            for _ in range(2):
                patch = ImageEnhance.Brightness(patch).enhance(random.choice([low, high]))
        return patch

    def __getitem__(self, index: int):
        original = Image.open(self.images_filenames[index]).resize(self.imsize).convert('RGB')
        y = random.randint(0, 3)
        x = original.copy()
        if not self.patch_localization and self.subject not in constants.NON_FIXED_OBJECTS():
            x = RandomAffine(3, scale=(1.05, 1.1))(x)
        cutting = random.choice(self.images_for_cut) if self.subject in constants.TEXTURES() else original
        seg = obj_mask(original) if self.subject in constants.NON_FIXED_OBJECTS() else self.fixed_segmentation
        k_patch, k_scar = 1.75, 2
        if self.patch_localization:
            if self.subject == 'capsule':
                x, seg = x.crop((0, 50, 255, 200)), seg.crop((0, 50, 255, 200))
            if self.subject == 'screw':
                x, seg = x.crop((25, 25, 230, 230)), seg.crop((25, 25, 230, 230))
            ps = self.patch_size
            left, top = random.randint(0, x.size[0] - ps), random.randint(0, x.size[1] - ps)
            box = (left, top, left + ps, top + ps)
            x, seg = x.crop(box), seg.crop(box)
            cutting = RandomCrop(ps)(cutting)
            k_patch = k_scar = 1
            if torch.sum(to_tensor(seg)) < int((ps * ps) / 2):
                y = 0
        if y > 0:
            binary = np.array(seg.convert('1'))
            coords_map = np.flip(np.column_stack(np.where(binary == 1)), axis=1)       # (x, y) pairs
            if y == 1:                                                                  # polygon patch
                centre = get_random_coordinate(coords_map)
                patch = self._decorrelate(x, self._defect_source(cutting, self.patch_area_ratio, CPP.rectangle_aspect_ratio))
                at = check_valid_coordinates_by_container(x.size, patch.size, current_coords=centre,
                                                          container_scaling_factor=k_patch)
                x = paste_patch(x, patch, at, rect2poly(patch, regular=False, sides=8))
            elif y == 2:                                                                # rotated scars
                scar = self._decorrelate(x, self._defect_source(cutting, self.scar_area_ratio, CPP.scar_aspect_ratio))
                scar = scar.convert('RGBA')
                copies, angle = random.randint(2, 5), random.randint(-45, 45)
                s = scar.rotate(angle, expand=True)
                for _ in range(copies):
                    centre = get_random_coordinate(coords_map)
                    at = check_valid_coordinates_by_container(x.size, s.size, current_coords=centre,
                                                              container_scaling_factor=k_scar)
                    x = paste_patch(x, s, at, s)
            else:                                                                       # poly-line
                draw = ImageDraw.Draw(x)
                side = random.choice(['left', 'top'])
                n = 30 if self.patch_localization else 60
                pts, c = [], 0
                for i in range(n):
                    idx = random.randint(c, int(len(coords_map) * (i / n)))
                    pts.append(tuple(coords_map[idx]))
                    c = idx
                rgb = random.choice(['black', 'white', 'silver'])
                if side == 'left':
                    pts.sort(key=lambda t: t[0])
                pts = savgol_filter(pts, 10, 2, axis=0)
                if not self.patch_localization:
                    pts = np.array_split(pts, 10)[random.randint(0, 9)]
                draw.line([tuple(p) for p in pts], fill=rgb, width=1 if self.patch_localization else 3)
        x = CPP.jitter_transforms(x)
        if self.transform:
            x = self.transform(x)
        return x, y, to_tensor(original)

    def __len__(self):
        return len(self.images_filenames)


class PretextTaskDatamodule(_DataModule):
    def __init__(self, subject: str, root_dir: str, imsize: tuple = (256, 256), batch_size: int = 32,
                 train_val_split: float = 0.2, seed: int = 0, min_dataset_length: int = 1000, duplication: bool = True,
                 patch_localization: bool = False, patch_size: tuple = 64, dataset_root: str = None,
                 swap_train_val: bool = True, gpu_pipeline: bool = False):
        self._gpu_loaders = {}
        self.root_dir_train, self.root_dir_test = root_dir + '/train/good/', root_dir + '/test/good/'
        self.subject, self.imsize, self.batch_size, self.train_val_split = subject, imsize, batch_size, train_val_split
        self.seed, self.min_dataset_length, self.duplication = seed, min_dataset_length, duplication
        self.patch_localization, self.patch_size = patch_localization, patch_size
        # the reference's Dataset reads sibling categories from a hard-coded 'dataset/' (datasets.py:189-200)
        self.dataset_root = dataset_root if dataset_root is not None else os.path.dirname(os.path.normpath(root_dir)) + '/'
        self.swap_train_val = swap_train_val      # quirk Q1: setup() builds train from the val names and vice versa
        self.gpu_pipeline = gpu_pipeline          # synthesise training batches on the GPU (augment.GpuPretextLoader)
        self.transform = _default_transform()
        self.prepare_filenames()

    def prepare_filenames(self):
        from sklearn.model_selection import train_test_split as tts
        names = get_filenames(self.root_dir_train)
        tr, va = tts(names, test_size=self.train_val_split, random_state=self.seed)
        te = get_filenames(self.root_dir_test)
        if self.duplication:
            tr, va, te = (duplicate_filenames(n, self.min_dataset_length) for n in (tr, va, te))
        self.train_images_filenames, self.val_images_filenames, self.test_images_filenames = tr, va, te
        # quirk Q2: the reference shuffles temporaries (datasets.py:464-466), i.e. the order is left untouched

    def _ds(self, names):
        return PretextTaskDataset(self.subject, names, imsize=self.imsize, transform=self.transform,
                                  patch_localization=self.patch_localization, patch_size=self.patch_size,
                                  dataset_root=self.dataset_root)

    def setup(self, stage: str = None) -> None:
        if stage == 'fit' or stage is None:
            a, b = self.train_images_filenames, self.val_images_filenames
            if self.swap_train_val:
                a, b = b, a
            self.train_dataset, self.val_dataset = self._ds(a), self._ds(b)
        if stage in ('test', 'predict') or stage is None:
            self.test_dataset = self._ds(self.test_images_filenames)

    def _gpu_loader(self, which, dataset, shuffle, seed):
        """One GPU-resident loader per split for the life of the datamodule: the category is decoded / uploaded once, and a second
        ``fit`` on the same datamodule (tools.training: projection head, then fine tuning) continues the loader's random stream
        instead of replaying it (GpuPretextLoader.stage)."""
        ld = self._gpu_loaders.get(which)
        if ld is None or ld.dataset is not dataset:
            from .augment import GpuPretextLoader
            ld = self._gpu_loaders[which] = GpuPretextLoader(dataset, self.batch_size, shuffle=shuffle, drop_last=True,
                                                             num_workers=min(self.num_workers, os.cpu_count() or 1), base_seed=seed)
        return ld

    def train_dataloader(self):
        if self.gpu_pipeline:
            return self._gpu_loader("train", self.train_dataset, True, self.seed)
        return self._loader(self.train_dataset, True, drop_last=True)

    def val_dataloader(self):
        if self.gpu_pipeline:
            return self._gpu_loader("val", self.val_dataset, False, self.seed + 1)
        return self._loader(self.val_dataset, False, drop_last=True)

    def test_dataloader(self):
        return self._loader(self.test_dataset, False)

    def predict_dataloader(self):
        return self._loader(self.test_dataset, False)
