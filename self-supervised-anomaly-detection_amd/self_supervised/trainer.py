"""Minimal fit / predict loop honouring the hooks the reference relies on from pytorch_lightning.Trainer
(src/self_supervised/tools.py:260-304, :327-347): ``model.train()`` at fit start (quirk Q6), ``training_step`` per
batch, ``on_train_epoch_end``, ``validation_step``, per-epoch scheduler step, callbacks with ``logged_metrics``,
``save_checkpoint`` (with ``on_save_checkpoint``), ``predict`` returning the list of ``predict_step`` outputs.

One process per GPU: when ``torch.distributed`` is initialised the train set is sharded with a DistributedSampler
and gradients are all-reduced by training.DataParallelStep (RCCL); the reference itself is single-device."""
import os
import time

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from . import training


class Callback:
    def on_train_epoch_end(self, trainer, pl_module):
        pass


class MetricTracker(Callback):
    """Per-epoch accuracy / loss history (mirrors src/self_supervised/custom_callbacks.py:5-24)."""

    def __init__(self):
        self.log_metrics = {'train': {'accuracy': [], 'loss': []}, 'val': {'accuracy': [], 'loss': []}}

    def on_train_epoch_end(self, trainer, pl_module):
        logs = trainer.logged_metrics
        for split in ('train', 'val'):
            for k in ('accuracy', 'loss'):
                if f'{split}_{k}' in logs:
                    self.log_metrics[split][k].append(float(logs[f'{split}_{k}']))


class ModelCheckpoint(Callback):
    """save_top_k=1 on a monitored metric, checked every ``every_n_epochs`` (tools.py:284-290)."""

    def __init__(self, dirpath, filename='best', save_top_k=1, monitor='val_loss', mode='min', every_n_epochs=1):
        self.dirpath, self.filename, self.monitor, self.mode, self.every = dirpath, filename, monitor, mode, every_n_epochs
        self.best, self.best_model_path = None, None

    def on_train_epoch_end(self, trainer, pl_module):
        if (trainer.current_epoch + 1) % self.every or self.monitor not in trainer.logged_metrics or trainer.global_rank:
            return
        v = float(trainer.logged_metrics[self.monitor])
        better = self.best is None or (v < self.best if self.mode == 'min' else v > self.best)
        if better:
            self.best = v
            os.makedirs(self.dirpath, exist_ok=True)
            self.best_model_path = os.path.join(self.dirpath, self.filename + '.ckpt')
            trainer.save_checkpoint(self.best_model_path, wait=False)      # written in the background, joined at the end of fit


def _to_device(batch, dev):
    """Host batch -> device.  (Round 3: staging through pinned buffers was tried against the stalls described at
    datasets._worker_context -- it did not remove them, the fork server does -- and writing 786 KB into pinned memory took 1.6 ms
    against 0.1 ms for the pageable copy, so the copy stays direct.)"""
    return tuple(b.to(dev, non_blocking=True) if torch.is_tensor(b) else b for b in batch)


class Trainer:
    def __init__(self, default_root_dir=None, callbacks=None, precision=32, benchmark=False, accelerator='auto',
                 devices=1, max_epochs=1, check_val_every_n_epoch=1, limit_train_batches=None, limit_val_batches=None,
                 enable_progress=False):
        self.default_root_dir, self.callbacks, self.max_epochs = default_root_dir, list(callbacks or []), max_epochs
        # 32: exact fp32 MFMA; 16 (what the reference passes = fp16 autocast): fp16-operand MFMA + dynamic loss scaling;
        # 'bf16': bf16-operand MFMA (explicit opt-in); see training.precision_mode
        self.precision = precision
        self.check_val_every_n_epoch = check_val_every_n_epoch
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.logged_metrics, self.current_epoch, self.global_step, self.model = {}, 0, 0, None
        self.global_rank, self.world = world_info()
        if not torch.cuda.is_available():
            raise RuntimeError("Trainer drives the MI355X HIP kernels: no GPU visible (there is no CPU fallback)")
        # one rank per GPU; more ranks than GPUs (rehearsing N ranks on a one-GPU box over gloo) share device 0, 1, ...
        self.device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1))

    def _shard(self, loader, epoch):
        if hasattr(loader, "shard"):                    # GPU-resident loader: shards / reshuffles itself
            return loader.shard(self.world, self.global_rank, epoch)
        if self.world == 1:
            return loader
        sampler = DistributedSampler(loader.dataset, num_replicas=self.world, rank=self.global_rank, shuffle=True)
        sampler.set_epoch(epoch)
        # the same worker start-up rule as the single-rank loaders (datasets._worker_context: a fork server once the GPU is up,
        # `__main__` hidden from the workers' bootstrap)
        from .datasets import _Loader, _worker_context
        return _Loader(loader.dataset, batch_size=loader.batch_size, sampler=sampler, drop_last=True,
                       num_workers=loader.num_workers,
                       multiprocessing_context=_worker_context() if loader.num_workers else None)

    def fit(self, model, datamodule=None, train_dataloaders=None, val_dataloaders=None):
        self.model = model
        model.trainer = self
        model.to(self.device)
        if datamodule is not None:
            if not hasattr(datamodule, 'train_dataset'):
                datamodule.setup('fit')
            train_dataloaders = train_dataloaders or datamodule.train_dataloader()
            val_dataloaders = val_dataloaders or datamodule.val_dataloader()
        model.train()                                   # PL puts the whole module in train mode at fit start
        (opt,), scheds = model.configure_optimizers()
        step = training.DataParallelStep(model, lr=opt.param_groups[0]['lr'], momentum=opt.momentum,
                                         weight_decay=opt.weight_decay, world_size=self.world, precision=self.precision)
        step.opt = opt
        opt.grad_scale = 1.0 / self.world
        self.epoch_throughput = []                      # (images of this rank, seconds) per training epoch, validation excluded
        for epoch in range(self.max_epochs):
            self.current_epoch = model.current_epoch = epoch
            model.train()
            sums = torch.zeros(3, device=self.device)
            torch.cuda.synchronize(self.device)
            t_epoch, n_images = time.perf_counter(), 0
            bank_steps = []
            for i, batch in enumerate(self._shard(train_dataloaders, epoch)):
                if self.limit_train_batches is not None and i >= self.limit_train_batches:
                    break
                x, y, _ = _to_device(batch, self.device)
                if self.world > 1 and step.self_check_report is None:
                    # first batch of a multi-rank fit: eager step == replayed step, replicas identical (training.self_check);
                    # falls back to eager launches, collectively, when they are not
                    rep = step.self_check(x.contiguous().float(), y.contiguous())
                    self.self_check = rep
                    if self.global_rank == 0:
                        print(f">>> data-parallel self-check: {rep}")
                la = step.step(x.contiguous().float(), y.contiguous())
                n_images += int(x.shape[0])
                sums += torch.stack([la[0], la[1], torch.ones_like(la[0])])
                if epoch > int(self.max_epochs / 2):    # memory bank of well-classified normal samples (models.py:270-275)
                    # the reference moves the selected rows to the CPU every step (a device-to-host sync per step); here the
                    # step's embeddings and its selection mask stay on the device and the bank is brought up to date once, at
                    # the end of the epoch, in the same row order (step after step, rank after rank inside a step)
                    y_hat = torch.max(step.last_logits, 1).indices
                    bank_steps.append((step.last_embeddings.detach().clone(), (y == 0) & (y_hat == 0)))
                self.global_step += 1
            if bank_steps:
                model.memory_bank = torch.cat([model.memory_bank, gather_bank_steps(bank_steps).to('cpu')])
            model.on_train_epoch_end()
            torch.cuda.synchronize(self.device)
            self.epoch_throughput.append((n_images, time.perf_counter() - t_epoch))
            if self.world > 1:
                dist.all_reduce(sums)
            n = max(float(sums[2]), 1.0)
            self.logged_metrics['train_loss'], self.logged_metrics['train_accuracy'] = sums[0] / n, sums[1] / n
            if val_dataloaders is not None and (epoch + 1) % self.check_val_every_n_epoch == 0:
                vs = torch.zeros(3, device=self.device)
                if hasattr(val_dataloaders, "set_epoch"):      # GPU-resident loader: fresh synthetic samples every validation pass
                    val_dataloaders.set_epoch(epoch)
                for i, batch in enumerate(val_dataloaders):
                    if self.limit_val_batches is not None and i >= self.limit_val_batches:
                        break
                    m = model.validation_step(_to_device(batch, self.device), i)
                    vs += torch.stack([m['val_loss'], m['val_accuracy'], torch.ones_like(m['val_loss'])])
                n = max(float(vs[2]), 1.0)
                self.logged_metrics['val_loss'], self.logged_metrics['val_accuracy'] = vs[0] / n, vs[1] / n
            for cb in self.callbacks:
                cb.on_train_epoch_end(self, model)
            for s in scheds:
                s.step()
        self._join_save()
        model.eval()

    def predict(self, model, datamodule=None, dataloaders=None, shard=False, max_batches=None):
        """shard=True (multi-GPU scoring, SURVEY s.8e): this rank scores batches rank, rank + world, ... only -- images
        are independent, no data-path collective -- and returns its own list; `gather_in_order` rebuilds the full one."""
        self.model = model
        model.trainer = self
        model.to(self.device).eval()
        if dataloaders is None:
            if not hasattr(datamodule, 'test_dataset'):
                datamodule.setup('predict')
            dataloaders = datamodule.predict_dataloader()
        # The reference predicts with batch_size=1 (tools.py:336); the kernels want thousands of patches per launch.
        # Eval-mode outputs are per-sample independent, so up to `predict_group` images of consecutive batches run
        # through predict_step together and the result is split back: one container per ORIGINAL batch, as before.
        group = int(getattr(self, "predict_group", 16))
        outs, pend = [], []

        def flush():
            if not pend:
                return
            if len(pend) == 1:
                outs.append(model.predict_step(pend[0][1], pend[0][0]))
            else:
                sizes = [b[0].shape[0] for _, b in pend]
                merged = tuple(torch.cat([b[k] for _, b in pend]) for k in range(len(pend[0][1])))
                whole = model.predict_step(merged, pend[0][0])
                whole.to_cpu()          # nine device-to-host copies per group instead of nine per image; the split is host work
                per_image = whole.split(sum(sizes))
                o = 0
                for n in sizes:
                    c = type(per_image[0])()
                    c.from_list(per_image[o:o + n])
                    outs.append(c)
                    o += n
            pend.clear()

        def mergeable(a, b):
            return (isinstance(a, (tuple, list)) and len(a) == len(b) and
                    all(torch.is_tensor(u) and torch.is_tensor(v) and u.shape[1:] == v.shape[1:] and u.dtype == v.dtype
                        for u, v in zip(a, b)))

        with torch.no_grad():
            for i, batch in enumerate(dataloaders):
                if max_batches is not None and i >= max_batches:
                    break               # the caller only uses the first batches (tools.inference's normality image)
                if shard and i % self.world != self.global_rank:
                    continue
                batch = _to_device(batch, self.device)
                if pend and (not mergeable(pend[-1][1], batch) or sum(b[0].shape[0] for _, b in pend) >= group):
                    flush()
                if group <= 1 or not isinstance(batch, (tuple, list)) or not all(torch.is_tensor(t) for t in batch):
                    flush()
                    outs.append(model.predict_step(batch, i))
                else:
                    pend.append((i, batch))
            flush()
        return outs

    def save_checkpoint(self, path, weights_only=False, wait=True):
        """``Trainer.save_checkpoint`` of Lightning (same keys).  The state leaves the device as ONE copy per dtype (a device-side
        concatenation of the ~150 tensors, one transfer, per-tensor clones on the host -- not one synchronising copy per tensor:
        147 -> ~30 ms per checkpoint, tools/profile_training.py).  wait=False (ModelCheckpoint inside ``fit``): the file is written
        by a background thread that the next save and the end of ``fit`` join -- nobody reads 'best so far' before that."""
        if self.global_rank:
            return
        self._join_save()
        m = self.model
        sd = m.state_dict()
        host, groups = {}, {}
        for k, v in sd.items():
            v = v.detach()
            if v.device.type == 'cpu':
                host[k] = v.contiguous().clone()
            else:
                groups.setdefault((v.dtype, v.device), []).append((k, v))
        for items in groups.values():
            flat = torch.cat([v.reshape(-1) for _, v in items]).cpu()
            off = 0
            for k, v in items:
                host[k] = flat[off:off + v.numel()].clone().view(v.shape)
                off += v.numel()
        ck = {'epoch': self.current_epoch, 'global_step': self.global_step, 'pytorch-lightning_version': '1.9.0',
              'state_dict': {k: host[k] for k in sd}, 'hyper_parameters': dict(m.hparams)}
        m.on_save_checkpoint(ck)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        if wait:
            torch.save(ck, path)
            return
        import threading
        err = []

        def write():
            try:
                torch.save(ck, path)
            except BaseException as e:     # noqa: BLE001  (re-raised by _join_save on the training thread)
                err.append(e)
        t = threading.Thread(target=write, name="ssad-checkpoint-writer")
        t.start()
        self._save_thread = (t, err)

    def _join_save(self):
        """Wait for the checkpoint a ModelCheckpoint callback left with the writer thread; its error, if any, is raised here."""
        pending = getattr(self, "_save_thread", None)
        if pending is not None:
            self._save_thread = None
            pending[0].join()
            if pending[1]:
                raise pending[1][0]


# ---------------------------------------------------------------------------------------------
# multi-GPU scoring helpers (one process per GPU; units = images, round-robin over ranks)
# ---------------------------------------------------------------------------------------------
_LOCAL_ONLY = 0


class local_only:
    """Context manager: inside it this process behaves as a single-GPU job (Trainer, tools.inference) even though
    torch.distributed is initialised -- a rank working on its own category of a category-parallel sweep."""

    def __enter__(self):
        global _LOCAL_ONLY
        _LOCAL_ONLY += 1

    def __exit__(self, *exc):
        global _LOCAL_ONLY
        _LOCAL_ONLY -= 1


def world_info():
    """(rank, world) of the default process group; (0, 1) when torch.distributed is not initialised or inside
    `local_only()`."""
    if not _LOCAL_ONLY and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def barrier():
    """dist.barrier() under torch.distributed, nothing otherwise (or inside `local_only()`)."""
    if world_info()[1] > 1:
        dist.barrier()


def gather_bank_rows(embeddings, mask):
    """Rows of `embeddings` [B][D] selected by `mask` [B] from EVERY rank, concatenated in rank order (SURVEY s.8e: the
    memory bank of models.py:270-275 under data parallelism).  Two fixed-shape exchanges: the int64 (batch size, row count)
    pair of every rank, then the rows padded to the largest batch size; ranks may hold different batch sizes (a ragged last
    batch).  The result is assembled on the device with ONE host read (the counts); with one rank this is plain indexing."""
    rank, world = world_info()
    sel = embeddings[mask].detach()
    if world == 1:
        return sel
    b, d = embeddings.shape
    mine = torch.tensor([b, sel.shape[0]], device=embeddings.device, dtype=torch.int64)
    meta = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(meta, mine)
    meta = torch.stack(meta).cpu()                       # the one synchronising read
    bmax = int(meta[:, 0].max())
    pad = torch.zeros((bmax, d), device=embeddings.device, dtype=embeddings.dtype)
    pad[:sel.shape[0]] = sel
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:int(n)] for p, n in zip(parts, meta[:, 1])])


def gather_bank_steps(steps):
    """steps: [(embeddings [B][D], mask [B])] of this rank's training steps.  Returns the selected rows of every rank in the order
    per-step gathering gives -- step 0 of rank 0, step 0 of rank 1, ..., step 1 of rank 0, ... -- with ONE exchange per epoch
    instead of one per step.  Batches may differ in size (a user's loader without drop_last, ranks with different last batches):
    every step is padded to the largest batch of any rank with rows whose mask is off; the mask travels as uint8."""
    rank, world = world_info()
    dev, d = steps[0][0].device, steps[0][0].shape[-1]
    sizes = [int(e.shape[0]) for e, _ in steps]
    bmax = max(sizes)
    if world > 1:
        t = torch.tensor([bmax], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bmax = int(t.item())
    if all(b == bmax for b in sizes):
        emb = torch.stack([e for e, _ in steps])                     # [S][B][D]
        msk = torch.stack([m for _, m in steps]).to(torch.uint8)     # [S][B]
    else:
        emb = torch.zeros((len(steps), bmax, d), device=dev, dtype=steps[0][0].dtype)
        msk = torch.zeros((len(steps), bmax), device=dev, dtype=torch.uint8)
        for i, (e, m) in enumerate(steps):
            emb[i, :e.shape[0]] = e
            msk[i, :e.shape[0]] = m.to(torch.uint8)
    if world > 1:
        es = [torch.empty_like(emb) for _ in range(world)]
        ms = [torch.empty_like(msk) for _ in range(world)]
        dist.all_gather(es, emb)
        dist.all_gather(ms, msk)
        emb = torch.stack(es, dim=1)                             # [S][W][B][D]
        msk = torch.stack(ms, dim=1)
    return emb.reshape(-1, emb.shape[-1])[msk.reshape(-1).bool()]


def gather_in_order(local_items, total):
    """Every rank holds the items of global indices rank, rank + world, ... (in that order); returns the full list in
    global order on every rank.  One all_gather_object at the END of scoring -- the only exchange of the scoring path."""
    rank, world = world_info()
    if world == 1:
        assert len(local_items) == total
        return list(local_items)
    parts = [None] * world
    dist.all_gather_object(parts, list(local_items))
    out = []
    for i in range(total):
        out.append(parts[i % world][i // world])
    return out


def broadcast_bank(obj, src=0):
    """The normality bank (<= 1000 x 512 fp32, plus its threshold) is fitted on one rank -- the 70/30 split draws from
    the global numpy RNG -- and sent to the others once.  `obj` is any picklable object on `src`, ignored elsewhere."""
    rank, world = world_info()
    if world == 1:
        return obj
    box = [obj if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]
