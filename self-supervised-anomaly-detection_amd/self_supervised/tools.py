"""Pipelines with the reference's signatures (src/self_supervised/tools.py:52-146, :204-399).

``training`` = the two-stage schedule (head only, then the whole net), ``inference`` = predict + cosine 3-NN
scoring, ``upsample`` = the fused blur+ReLU+bilinear HIP kernel, ``Evaluator`` = AUROC / F1 / AUPRO / IoU.
They drive trainer.Trainer (PyTorch Lightning is not required).  Plots (the reference's ``vis.*`` calls) are out of
scope: the histories / curves are returned or written as JSON instead."""
import json
import os
import random
import shutil

import numpy as np
import torch
from torch import Tensor

from . import metrics as mtr
from . import ops
from .constants import METRICS, EvaluationOutputContainer, ModelOutputsContainer
from .datasets import MVTecDatamodule, PretextTaskDatamodule
from .models import AnomalyDetector, PeraNet
from .trainer import MetricTracker, ModelCheckpoint, Trainer, barrier, broadcast_bank, gather_in_order, local_only, world_info


class Evaluator:
    """tools.py:52-146: scores an inference output; pixel level when ``patch_level`` (maps vs ground truths)."""

    def __init__(self, evaluation_metrics: list = []) -> None:
        self.evaluation_metrics = np.array(evaluation_metrics)
        self.scores = EvaluationOutputContainer()
        self.curves = {}

    def evaluate(self, output_container: ModelOutputsContainer, subject: str, outputs_dir: str, patch_level: bool = False):
        print('>>> start evaluation')
        if self.evaluation_metrics.size == 0:
            print('No metrics selected')
        bad = [m for m in self.evaluation_metrics if m not in METRICS()]
        if bad:
            raise ValueError(f"Wrong metric(s): {bad}; correct metrics list: {METRICS()}")
        targets, scores = output_container.y_true_binary_labels, output_container.anomaly_maps
        if patch_level:
            targets = torch.flatten(output_container.ground_truths, 0, -1)
            scores = torch.flatten(scores, 0, -1)
        on_gpu = scores.is_cuda
        for m in self.evaluation_metrics:
            if m != 'auroc' and ((m == 'f1-score') == patch_level):
                raise ValueError(f"'{m}' not a valid metric for '{'patch' if patch_level else 'image'}-level' mode")
        if on_gpu:
            # maps still on the device (straight from tools.upsample): every sort-bound metric runs there (csrc/auroc.hip) --
            # on the host the three argsorts of a category's 6 M pixel scores take longer than training the category
            targets_dev = targets.to(scores.device)
            threshold = self._get_threshold(scores, targets_dev)         # one overridable hook for host and device maps
            if 'auroc' in self.evaluation_metrics:
                print(' pixel auroc' if patch_level else '>>> image auroc')
                self.scores.auroc = mtr.auroc_gpu(targets_dev, scores)
            if 'f1-score' in self.evaluation_metrics:
                self.scores.f1_score = mtr.compute_f1_gpu(targets_dev, scores, threshold)
            if 'aupro' in self.evaluation_metrics:
                fprs, pros = mtr.compute_pro_gpu(output_container.anomaly_maps.squeeze(1), output_container.ground_truths.squeeze(1))
                self.scores.aupro = mtr.compute_aupro(fprs, pros, 0.3)
                self.curves['pro'] = (fprs, pros)
            if 'iou' in self.evaluation_metrics:
                self.scores.iou = mtr.compute_iou_gpu(scores, targets_dev, threshold)
        else:
            targets, scores = targets.detach().cpu(), scores.detach().cpu()
            threshold = self._get_threshold(scores, targets)
            if 'auroc' in self.evaluation_metrics:
                print(' pixel auroc' if patch_level else '>>> image auroc')
                fpr, tpr, _ = mtr.compute_roc(targets, scores)
                self.scores.auroc = mtr.compute_auc(fpr, tpr)
                self.curves['roc'] = (fpr, tpr)
            if 'f1-score' in self.evaluation_metrics:
                self.scores.f1_score = mtr.compute_f1(targets, scores, threshold)
            if 'aupro' in self.evaluation_metrics:
                fprs, pros = mtr.compute_pro(output_container.anomaly_maps.detach().cpu().squeeze(1).numpy(),
                                             output_container.ground_truths.detach().cpu().squeeze(1).numpy())
                self.scores.aupro = mtr.compute_aupro(fprs, pros, 0.3)
                self.curves['pro'] = (fprs, pros)
            if 'iou' in self.evaluation_metrics:
                self.scores.iou = mtr.compute_iou(scores, targets, threshold)
        if outputs_dir:
            os.makedirs(outputs_dir, exist_ok=True)
            with open(os.path.join(outputs_dir, subject + ('_pixel' if patch_level else '_image') + '_scores.json'), 'w') as f:
                json.dump({k: (None if v is None else float(v)) for k, v in vars(self.scores).items()}, f)

    def _get_threshold(self, scores: Tensor, targets: Tensor):
        """F1-optimal threshold (tools.py:131-146 of the reference); device tensors take the device kernels, same value.  A subclass
        that overrides this is honoured on both routes.  (curves['roc'] is only filled for host maps: the device route returns the
        area, not the curve.)"""
        if scores.is_cuda:
            return mtr.best_f1_threshold_gpu(scores, targets)
        return mtr.best_f1_threshold(scores, targets)


def _seed_everything(seed):
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)


def training(dataset_dir: str, outputs_dir: str, subject: str, imsize: tuple = (256, 256), patch_localization: bool = False,
             patchsize: int = 32, seed: int = 0, batch_size: int = 96, projection_training_params: tuple = (10, 0.03),
             fine_tune_params: tuple = (30, 0.005), trainer_kwargs: dict = None, gpu_pipeline: bool = None) -> dict:
    """tools.py:204-306.  Returns the two metric histories (the reference plots them).
    gpu_pipeline: True = batches synthesised on the GPU from host-drawn parameter records (the sampler restates __getitem__'s draws,
    the kernels are byte-identical to the PIL path per record; batches are seeded one by one; 5.7 k img/s fp32 / 8.1 k precision 16
    at batch 96); False = the reference's input path (PretextTaskDataset.__getitem__ in 8 DataLoader workers, ~300 img/s at
    256 x 256); None (default) = SSAD_GPU_PIPELINE from the environment when set, else True: the training loop needs the GPU anyway,
    the two paths draw from the same distributions, and neither reproduces the reference's stream sample for sample (its workers
    are seeded from torch's base seed per epoch)."""
    if gpu_pipeline is None:
        gpu_pipeline = os.environ.get("SSAD_GPU_PIPELINE", "1") != "0"
    print('>>> initializing training')
    checkpoint_name = 'best_model.ckpt'
    proj_epochs, proj_lr = projection_training_params
    fine_tune_epochs, fine_tune_lr = fine_tune_params
    # under torch.distributed (one process per GPU; the reference is single-device) the filesystem side belongs to rank 0
    # and every hand-over through a file is fenced by a barrier
    rank, _ = world_info()
    if rank == 0:
        os.makedirs(outputs_dir, exist_ok=True)
        if os.path.exists(outputs_dir + 'logs/'):
            shutil.rmtree(outputs_dir + 'logs/')
    barrier()
    print('>>> setting seeds')
    _seed_everything(seed)
    print('>>> preparing datamodule')
    datamodule = PretextTaskDatamodule(subject, dataset_dir, imsize=imsize, batch_size=batch_size, seed=seed,
                                       patch_localization=patch_localization, patch_size=patchsize, gpu_pipeline=gpu_pipeline)
    datamodule.setup()
    if world_info()[1] > 1:      # SURVEY s.8e: each rank draws its own augmentations (rank-offset seed); the split above is shared
        random.seed(seed + rank)
        np.random.seed(seed + rank)
    tk = dict(trainer_kwargs or {})
    print('>>> preparing model')
    pretext_model = PeraNet(learning_rate=proj_lr, epochs=proj_epochs)
    pretext_model.freeze_net(['backbone'])
    cb = MetricTracker()
    # the reference's Trainer arguments (tools.py:260-267); trainer_kwargs may override them (e.g. precision=32)
    targs = dict(default_root_dir=outputs_dir + 'logs/', precision=16, benchmark=True, accelerator='auto', devices=1,
                 check_val_every_n_epoch=1)
    targs.update(tk)
    trainer = Trainer(callbacks=[cb], max_epochs=proj_epochs, **targs)
    print('>>> training projection head')
    trainer.fit(pretext_model, datamodule=datamodule)
    history = {'projection_train': cb.log_metrics}
    throughput = {'projection_train': list(trainer.epoch_throughput)}
    pretext_model.clear_memory_bank()
    trainer.save_checkpoint(outputs_dir + checkpoint_name, weights_only=True)     # rank 0 writes ...
    barrier()                                                                      # ... before anybody reads

    print('>>> setting up the model (fine tune whole net)')
    pretext_model = PeraNet.load_from_checkpoint(outputs_dir + checkpoint_name, learning_rate=fine_tune_lr,
                                                 epochs=fine_tune_epochs, stage='fine_tune')
    pretext_model.unfreeze()
    mc = ModelCheckpoint(dirpath=outputs_dir + 'logs/', filename='best_model_so_far', save_top_k=1, monitor="val_loss",
                         mode='min', every_n_epochs=5)
    cb = MetricTracker()
    trainer = Trainer(callbacks=[cb, mc], max_epochs=fine_tune_epochs, **targs)
    print('>>> Fine tuning')
    trainer.fit(pretext_model, datamodule=datamodule)
    trainer.save_checkpoint(outputs_dir + checkpoint_name)
    history['fine_tune'] = cb.log_metrics
    throughput['fine_tune'] = list(trainer.epoch_throughput)
    history['throughput'] = throughput          # (images, seconds) per training epoch of this rank: data feeding included
    if rank == 0:
        with open(outputs_dir + 'history.json', 'w') as f:
            json.dump(history, f)
    barrier()
    return history


def _fast_mvtec_ok(dataset):
    """The streamed predict below restates MVTecDataset.__getitem__ for the default transform only."""
    from .datasets import IMAGENET_MEAN, IMAGENET_STD, MVTecDataset
    from .tv_transforms import Compose, Normalize, ToTensor
    t = getattr(dataset, "transform", None)
    return (type(dataset) is MVTecDataset and isinstance(t, Compose) and len(t.ts) == 2 and isinstance(t.ts[0], ToTensor)
            and isinstance(t.ts[1], Normalize) and t.ts[1].mean.flatten().tolist() == torch.tensor(IMAGENET_MEAN).tolist()
            and t.ts[1].std.flatten().tolist() == torch.tensor(IMAGENET_STD).tolist()
            and os.environ.get("SSAD_FAST_PREDICT", "1") != "0")


def _decode_threads():
    """Decode threads for the streamed predict: the CPUs this process may actually use (the cgroup quota where one is set: a 1-GPU job
    on a 256-thread host is given 16), at most 16 -- Pillow's PNG inflate runs outside the GIL, so the threads scale until the quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(16, n, int(os.environ.get("SSAD_DECODE_THREADS", "16"))))


class _MVTecPrefetch:
    """The host half of the streamed predict, startable before the model exists: a thread pool decodes the test files (Pillow decodes
    outside the GIL) at their native size and reads the ground-truth masks exactly as ``MVTecDataset.__getitem__`` does
    (``get_ground_truth``: resize + dither to mode '1' stay Pillow's).  tools.inference starts it before it loads the checkpoint."""

    def __init__(self, dataset, indices, threads: int = None, extra_files=()):
        from concurrent.futures import ThreadPoolExecutor
        from . import gpu_io
        from .functional import get_ground_truth, get_ground_truth_filename
        self.dataset, self.indices = dataset, list(indices)
        # extra_files: images decoded and scored IN FRONT of the dataset's own (tools.inference: the training image whose embeddings
        # become the normality bank rides in the first group instead of a DataLoader pass of its own); 'good' images: all-zero masks
        self.extra = list(extra_files)
        n = len(self.extra) + len(self.indices)
        w_img, h_img = dataset.imsize
        names = self.extra + [dataset.images_filenames[i] for i in self.indices]
        self.gt8 = np.zeros((n, h_img, w_img), np.uint8)
        self.native = [None] * n
        gt_dir = dataset.dataset_dir + 'ground_truth/'

        def load(j):
            self.native[j] = gpu_io.read_native(names[j])   # decoded at its native size; the resize runs on the device (gpu_io)
            gfile = get_ground_truth_filename(names[j], gt_dir) if j >= len(self.extra) else None
            if gfile:                                        # 'good' images: Image.new(mode='1') = all zeros
                self.gt8[j] = np.asarray(get_ground_truth(gfile, dataset.imsize).convert('L'))
        self.pool = ThreadPoolExecutor(threads or _decode_threads())
        self._load, self.futs, self._next = load, [None] * n, 0
        # a sliding window of decodes ahead of the consumer (a real MVTec category is 100-170 test images of 1024 x 1024: decoding them
        # all at once would hold 0.3-0.5 GB of native-size arrays with no back-pressure from the GPU side)
        self.window = int(os.environ.get("SSAD_PREFETCH_WINDOW", "96"))
        self._submit_until(self.window)

    def _submit_until(self, end):
        end = min(end, len(self.futs))
        while self._next < end:
            self.futs[self._next] = self.pool.submit(self._load, self._next)
            self._next += 1

    def wait(self, a, b):
        self._submit_until(b + self.window)             # keep `window` images in flight beyond the group that is asked for
        for f in self.futs[a:b]:
            f.result()

    def close(self, cancel=False):
        self.pool.shutdown(wait=True, cancel_futures=cancel)


def _predict_mvtec_streamed(model: PeraNet, dataset, device, indices, group: int = 32, threads: int = None, prefetch=None):
    """``Trainer.predict`` + ``ModelOutputsContainer.from_list`` for an MVTecDataset with the default transform, as ONE stream
    instead of a DataLoader of batch size 1 (tools.py:336-347 of the reference): same values, field for field.

    * a thread pool (``_MVTecPrefetch``) decodes the files and reads the ground-truth masks;
    * ``group`` images at a time go to the device as uint8 at their NATIVE size, where Pillow's bicubic ``resize(imsize)`` (csrc/resize.hip,
      bit-exact), the 'L' -> 'RGB' replication and ToTensor + Normalize (two IEEE fp32 operations per value,
      ssad_u8hwc_to_f32chw_norm: bit-identical to the host transform) run, then ``model.forward`` -- thousands of patches per launch
      instead of 841;
    * results return to the host per group on a side stream while the next group computes; the embeddings also STAY on the device
      (second return value) for the detector, instead of making the round trip host -> device again.
    Returns (container with CPU tensors, device embeddings [n * P][D])."""
    import ctypes
    from . import _hip, gpu_io
    from .datasets import IMAGENET_MEAN, IMAGENET_STD
    from .functional import get_prediction_class
    from .converters import gt2label
    out = ModelOutputsContainer()
    pre = prefetch
    ne = len(pre.extra) if pre is not None else 0      # images in front of the dataset's own (their embeddings: out.extra_embeddings)
    n = ne + len(indices)
    if n == 0:
        if prefetch is not None:
            prefetch.close()
        return out, None
    w_img, h_img = dataset.imsize
    pre = prefetch if prefetch is not None else _MVTecPrefetch(dataset, indices, threads)
    assert pre.indices == list(indices)
    native, gt8 = pre.native, pre.gt8
    mean = (ctypes.c_float * 3)(*IMAGENET_MEAN)
    std = (ctypes.c_float * 3)(*IMAGENET_STD)
    orig = torch.empty((n, 3, h_img, w_img), dtype=torch.float32)
    xnorm = torch.empty((n, 3, h_img, w_img), dtype=torch.float32)
    emb_dev = logits_dev = emb_host = logits_host = None
    side = torch.cuda.Stream(device)
    main = torch.cuda.current_stream(device)
    lib = _hip.lib()
    pending = None                                       # (a, b, orig_dev, x_dev, event) of the group whose results are still on the device

    def drain(item):
        a, b, o_dev, x_dev, ev, p = item
        with torch.cuda.stream(side):
            side.wait_event(ev)
            orig[a:b].copy_(o_dev)
            xnorm[a:b].copy_(x_dev)
            emb_host[a * p:b * p].copy_(emb_dev[a * p:b * p])
            logits_host[a * p:b * p].copy_(logits_dev[a * p:b * p])
        for t in (o_dev, x_dev):
            t.record_stream(side)
    # groups of equal size, about `group` images each (97 images = 96 + the bank image: three forwards of 33 / 32 / 32, not 32 / 32 / 32
    # and a fourth one of a single image)
    ngroups = max(1, (n + group // 2) // group)
    per = -(-n // ngroups)
    try:
        with torch.no_grad():
            for a in range(0, n, per):
                b = min(n, a + per)
                pre.wait(a, b)
                img_dev = gpu_io.to_rgb_batch(native[a:b], dataset.imsize, device)
                native[a:b] = [None] * (b - a)
                o_dev = torch.empty((b - a, 3, h_img, w_img), device=device, dtype=torch.float32)
                x_dev = torch.empty_like(o_dev)
                _hip.check(lib.ssad_u8hwc_to_f32chw_norm(img_dev.data_ptr(), o_dev.data_ptr(), x_dev.data_ptr(), b - a, h_img, w_img,
                                                         mean, std, _hip.stream()))
                pred = model(x_dev)
                p = pred['latent_space'].shape[0] // (b - a)
                if emb_dev is None:
                    d, c = pred['latent_space'].shape[1], pred['classifier'].shape[1]
                    emb_dev = torch.empty((n * p, d), device=device, dtype=torch.float32)
                    logits_dev = torch.empty((n * p, c), device=device, dtype=torch.float32)
                    emb_host, logits_host = torch.empty((n * p, d)), torch.empty((n * p, c))
                emb_dev[a * p:b * p].copy_(pred['latent_space'])
                logits_dev[a * p:b * p].copy_(pred['classifier'])
                ev = torch.cuda.Event()
                ev.record(main)
                if pending is not None:
                    drain(pending)                          # the previous group's results travel while this group computes
                pending = (a, b, o_dev, x_dev, ev, p)
            drain(pending)
    except BaseException:
        pre.close(cancel=True)                           # a failing predict must not leave decode threads running until exit
        raise
    pre.close()
    side.synchronize()
    p = emb_dev.shape[0] // n
    if ne:
        # the images in front are not part of the dataset's output: their embeddings stay on the device for the caller
        out.extra_embeddings = emb_dev[:ne * p]
        orig, xnorm, gt8 = orig[ne:], xnorm[ne:], gt8[ne:]
        emb_host, logits_host, emb_dev = emb_host[ne * p:], logits_host[ne * p:], emb_dev[ne * p:]
    gts = torch.from_numpy(gt8).float().div_(255.0).unsqueeze(1)
    out.original_data, out.tensor_data, out.ground_truths = orig, xnorm, gts
    out.raw_predictions, out.embedding_vectors = logits_host, emb_host
    # (first maximum per row, as torch.max(x, 1).indices returns it -- functional.get_prediction_class -- through numpy: 1 ms
    # instead of 14 for 80 736 rows of four)
    out.y_hat = torch.from_numpy(logits_host.numpy().argmax(1))
    # predict_step labels every BATCH (of one image) from its ground truth (models.py:314-318)
    out.y_true_binary_labels = torch.tensor(gt2label(gts))
    out.y_true_multiclass_labels = torch.tensor(gt2label(gts, negative=-1, positive=model.num_classes))
    return out, emb_dev


TIMELINE = []       # SSAD_TIMELINE=1: (phase, perf_counter) marks of the last tools.inference call (tools/time_inference.py)


def _mark(name):
    if os.environ.get("SSAD_TIMELINE") == "1":
        import time
        TIMELINE.append((name, time.perf_counter()))


def inference(model_input_dir: str, dataset_dir: str, subject: str, mvtec_inference: bool = True,
              patch_localization: bool = False) -> ModelOutputsContainer:
    """tools.py:310-390."""
    del TIMELINE[:]
    print('>>> initializing inference')
    # MVTec test data: file lists and decode threads start BEFORE the checkpoint is read, so that the first group of images is
    # decoded by the time the model is on the device (the datamodule draws nothing from the global generators)
    rank, world = world_info()
    datamodule = prefetch = mine = None
    bank_file = bank_item = None
    if mvtec_inference:
        datamodule = MVTecDatamodule(dataset_dir, batch_size=1)
        datamodule.setup('predict')
        if _fast_mvtec_ok(datamodule.test_dataset):
            mine = list(range(rank, len(datamodule.test_dataset), world)) if world > 1 else list(range(len(datamodule.test_dataset)))
            # Which training image becomes the normality bank (tools.py:374-381: element [0] of a prediction over the SHUFFLED training
            # loader) is decided by two draws from torch's global generator -- the test loader's base seed, then the training loader's
            # base seed and its sampler's seed.  Nothing between here and there draws from it (the model is built on the meta device, the
            # forward pass uses no host generator), so the draws are made NOW, in the reference's order, and the image is decoded with
            # the test images and scored in their first group -- not by a DataLoader pass of its own after them (round 6: 95 ms of
            # host time per call).  Only the index is taken from the loader; the generator ends in the same state.
            try:
                state = torch.get_rng_state()
                torch.empty((), dtype=torch.int64).random_()                    # (the test loader's iterator, tools.py:336-347)
                ndm = MVTecDatamodule(dataset_dir, batch_size=1)
                ndm.num_workers = 0
                ndm.setup()
                it = iter(ndm.train_dataloader())
                first = it._next_index()                                         # the sampler's draw; no image is read
                bank_file = ndm.train_dataset.images_filenames[int(first[0])]
                bank_item = (ndm.train_dataset, int(first[0]))
                del it
            except Exception:                                                    # noqa: BLE001  (a DataLoader without these internals:
                torch.set_rng_state(state)                                       #  the draws are made later, by the loader itself)
                bank_file = None
            # (a rank without test images of its own -- more ranks than images -- scores nothing, the bank image included)
            prefetch = _MVTecPrefetch(datamodule.test_dataset, mine, extra_files=[bank_file] if (bank_file and mine) else ())
    print('>>> preparing model')
    _mark("prefetch-started")
    try:
        model = PeraNet.load_from_checkpoint(model_input_dir)
        _mark("checkpoint")
        model.eval()
        if patch_localization:
            model.enable_patch_level_mode()
        tester = Trainer(accelerator='auto', devices=1)
    except BaseException:
        if prefetch is not None:               # no model, no predict: the decode threads are not left behind
            prefetch.close(cancel=True)
        raise
    print('>>> preparing datamodule')
    if mvtec_inference:
        model.enable_mvtec_inference()
    else:
        datamodule = PretextTaskDatamodule(subject=subject, root_dir=dataset_dir, min_dataset_length=500, batch_size=1)
    print('>>> doing prediction')
    # under torch.distributed (one process per GPU) every rank scores its own round-robin share of the images; the
    # per-image containers are exchanged once at the end so that every rank returns the full output
    emb_dev = None
    if prefetch is not None:
        # the hot path of an evaluation: one stream through decode threads and large launches instead of a DataLoader of batch
        # size 1 (same values; _predict_mvtec_streamed).  The DataLoader iterator the reference creates here draws its base
        # seed from torch's global generator: the draw is kept, so that what follows (the shuffled loader of the normality
        # image) sees the same generator state
        if bank_file is None:
            torch.empty((), dtype=torch.int64).random_()
        model.to(tester.device).eval()
        _mark("model-on-device")
        output, emb_dev = _predict_mvtec_streamed(model, datamodule.test_dataset, tester.device, mine, prefetch=prefetch)
        _mark("predicted")
        n_pred = len(mine)
    else:
        predictions = tester.predict(model, datamodule=datamodule, shard=world > 1)
        output = ModelOutputsContainer()
        output.from_list(predictions)
        n_pred = len(predictions)
    print('>>> anomaly detection phase')
    if patch_localization:
        detector = AnomalyDetector(patch_level=True, batch=n_pred, num_patches=model.num_patches)
    else:
        detector = AnomalyDetector()
    if model.memory_bank.shape[0] > 1000:            # quirk Q3: the bank is capped at 1000 rows, so this never holds
        normality = model.memory_bank
    elif bank_file is not None and getattr(output, "extra_embeddings", None) is not None:
        print(' not enough data in memory bank, sampling new data trom train set')
        normality = output.extra_embeddings.cpu()       # the training image scored in front of the test images (see above)
        del output.extra_embeddings
    elif bank_item is not None:
        # the draws are made and the image is known, but it did not ride with test images (a rank that has none): score that very
        # image -- drawing again would pick another one
        print(' not enough data in memory bank, sampling new data trom train set')
        ds_, i_ = bank_item
        model.to(tester.device).eval()
        with torch.no_grad():
            one = model.predict_step(tuple(t.unsqueeze(0).to(tester.device) for t in ds_[i_]), 0)
        one.to_cpu()
        normality = one.embedding_vectors
    else:
        print(' not enough data in memory bank, sampling new data trom train set')
        if mvtec_inference:
            normality_datamodule = MVTecDatamodule(dataset_dir, batch_size=1)
        else:
            normality_datamodule = PretextTaskDatamodule(subject=subject, root_dir=dataset_dir, batch_size=1)
        if mvtec_inference:
            normality_datamodule.num_workers = 0      # one deterministic image is read: not worth eight worker processes
        normality_datamodule.setup()
        # the reference predicts the WHOLE training loader and keeps element [0] (tools.py:379-381); the first batch of a loader
        # does not depend on how far the loader is consumed afterwards (the shuffle permutation and the one draw for the workers'
        # base seed happen when iteration starts), so only that batch is predicted: same image, same RNG state, 1 / 209 of the work
        # (MVTec loader only: its __getitem__ draws no random numbers, so this also holds with in-process loading)
        output_normality = tester.predict(model, dataloaders=normality_datamodule.train_dataloader(),
                                          max_batches=1 if mvtec_inference else None)[0]
        output_normality.to_cpu()
        normality = output_normality.embedding_vectors
    output.to_cpu()
    if world > 1:
        # one bank for everybody: rank 0 draws the 70/30 split and fits, the others receive (bank, threshold)
        if rank == 0:
            detector.fit(normality)
        state = broadcast_bank((detector.bank.cpu(), detector.threshold) if rank == 0 else None)
        if rank != 0:
            detector.bank, detector.threshold = AnomalyDetector._dev(state[0]), state[1]
    else:
        detector.fit(normality)
    _mark("bank-fitted")
    print(' computing anomaly scores')
    output.anomaly_maps = detector.predict(emb_dev if emb_dev is not None else output.embedding_vectors).cpu()
    _mark("maps")
    if world > 1:
        n_total = len(datamodule.test_dataset)
        per_image = gather_in_order(_split_container(output, n_pred), n_total)
        output = ModelOutputsContainer()
        output.from_list(per_image)
    return output


def _split_container(output: ModelOutputsContainer, n_images: int):
    return output.split(n_images)


def upsample(anomaly_maps: Tensor, target_size: int = 256, verbose: bool = True):
    """tools.py:394-399: relu(gaussian_blur(k=7)) then bilinear to target_size, one fused kernel."""
    if verbose:
        print('>>> upsampling')
    m = torch.as_tensor(anomaly_maps, dtype=torch.float32)
    if not m.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("tools.upsample runs on the MI355X HIP kernel only (no CPU fallback)")
        m = m.cuda()
    return ops.blur_relu_bilinear(m.contiguous(), 7, target_size)


def gradcam_maps(model: PeraNet, images: Tensor, y_hat: Tensor, chunk: int = 64) -> Tensor:
    """Image-level localisation (src/evaluator.py:268-282 of the reference): a Grad-CAM saliency of the predicted class
    for every image predicted anomalous (y_hat != 0), an all-zero map otherwise; NaNs of constant maps become 0.
    images [N][3][H][H], y_hat [N] -> [N][1][H][H] on the model's device."""
    from .gradcam import GradCam
    cam = GradCam(model)
    dev = next(model.parameters()).device
    y_hat = torch.as_tensor(y_hat).reshape(-1).long()
    n, _, h, w = images.shape
    maps = torch.zeros((n, 1, h, w), device=dev, dtype=torch.float32)
    sel = torch.nonzero(y_hat != 0).reshape(-1)
    for i in range(0, sel.numel(), chunk):
        idx = sel[i:i + chunk]
        maps[idx.to(dev)] = cam(images[idx.to(images.device)], y_hat[idx])
    return torch.nan_to_num(maps)


def sweep(dataset_dir: str, outputs_dir: str, categories: list, imsize: tuple = (256, 256), patch_localization: bool = True,
          seed: int = 0, batch_size: int = 96, projection_training_params=(10, 0.03), fine_tune_params=(30, 0.005),
          metrics=('auroc', 'aupro', 'iou'), trainer_kwargs=None, tables_output: str = None, train: bool = True):
    """Category sweep (BASELINE configs[4]; the loop of src/evaluator.py:432-564 without its plots): per category
    training -> inference -> upsample -> Evaluator, one row of scores each plus an 'average' row, exported as csv /
    markdown when `tables_output` is given.  Categories are independent models: under torch.distributed (one process per
    GPU) rank r takes categories r, r + world, ... and the rows are exchanged once at the end -- no collective inside a
    category.  Returns the pandas DataFrame (identical on every rank)."""
    rank, world = world_info()
    mine = [c for i, c in enumerate(categories) if i % world == rank]
    rows = {}
    for subject in mine:
        sub_out = os.path.join(outputs_dir, subject) + '/'
        data = os.path.join(dataset_dir, subject) + '/'
        with local_only():          # a category is a single-GPU job: no sharding / all-reduce with ranks on other categories
            if train:
                training(data, sub_out, subject, imsize=imsize, patch_localization=patch_localization, seed=seed,
                         batch_size=batch_size, projection_training_params=projection_training_params,
                         fine_tune_params=fine_tune_params, trainer_kwargs=trainer_kwargs)
            out = inference(sub_out + 'best_model.ckpt', data, subject, mvtec_inference=True,
                            patch_localization=patch_localization)
        if patch_localization:
            out.anomaly_maps = upsample(out.anomaly_maps, int(out.ground_truths.shape[-1]), verbose=False)      # stays on the device: the Evaluator's GPU metrics
        ev = Evaluator(evaluation_metrics=[m for m in metrics if (m != 'f1-score') == patch_localization or m == 'auroc'])
        ev.evaluate(out, subject, sub_out, patch_level=patch_localization)
        rows[subject] = {k: v for k, v in vars(ev.scores).items() if v is not None}
    if world > 1:
        import torch.distributed as dist
        parts = [None] * world
        dist.all_gather_object(parts, rows)
        rows = {k: v for part in parts for k, v in part.items()}
    cols = sorted({k for r in rows.values() for k in r})
    table = {c: [float(rows[s].get(c, float('nan'))) for s in categories] for c in cols}
    for c in cols:
        table[c].append(float(np.nanmean(table[c])))
    df = mtr.metrics_to_dataframe(table, list(categories) + ['average'])
    if tables_output and rank == 0:
        name = 'patch_all_scores' if patch_localization else 'image_all_scores'
        mtr.export_dataframe(df, tables_output + 'csv/', name + '.csv')
        try:
            mtr.export_dataframe(df, tables_output + 'markdown/', name + '.md', mode='markdown')
        except ImportError:
            pass
    return df
