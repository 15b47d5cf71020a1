#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): train images/sec + anomaly-maps/sec, ResNet-18, 256x256, batch 256.

    python bench.py --gpus N --steps K --warmup W          # any N: for N > 1 this process only starts the N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # the same thing, launched by hand / by the driver

One "step" = one pass of the hot path over one batch of synthetic 256x256 images per GPU, inputs already resident in
HBM: (a) one pretext training step (forward + backward + SGD update, + RCCL gradient all-reduce when N > 1) and (b)
anomaly-map scoring of the same batch (841 sliding-window patches per image -> trunk -> 512-d embedding -> cosine
3-NN against a 588-row bank -> blur -> bilinear 256x256 map).  Both rates are reported; ``value`` is the training rate
(the first-named metric), ``anomaly_maps_per_sec`` the scoring rate.  fp32 throughout (exact f32 MFMA).

Partition (SURVEY s.8e): the headline is the STRONG partition the survey names -- global batch 256, 256 / N images per rank
for training and for scoring, "scaling": "strong" -- so the N = 1 line is the 256-image step and an N = 8 run is the same
job on eight ranks.  The same run also times WEAK scaling (256 images per rank) and reports it under "weak"; at N = 1 it
reports the per-rank work of the N = 8 partition (batch 32) under "batch32".  ``--scaling weak`` makes weak scaling the
headline instead (``--batch`` images per rank).

Profiling: put rocprofv3 on a single-rank command (`rocprofv3 ... -- python3 bench.py --gpus 1 ...`); with --gpus N > 1 this
process is only a launcher, so a profiler must wrap the per-rank command, never the launcher.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "self-supervised-anomaly-detection_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA (the 2:1-sparsity headline is twice that and never used here)
PEAK_HBM_GBPS = 8000.0
# SURVEY s.8d: algorithmic bytes of U_train with BN / ReLU / add fused into conv epilogues, every conv reading its input once and writing its
# output once: 3 x 33.55 MB per image with fp32 tensors (forward, input gradients, weight gradients) + 254 MB per step of optimizer
# traffic (12.69 M parameters: p, g, m read, p, m written).  With the tensors stored as halves (the precision-16 step) the activation
# part halves; parameters, their gradients and the optimizer stay fp32.
U_TRAIN_ACT_MB_F32 = 3 * 33.554432
OPTIMIZER_MB_PER_STEP = 253.83
U_TRAIN_GFLOP = 14.22            # SURVEY s.8d: one 256x256 image through fwd + bwd + update
U_MAP_GFLOP = 252.06             # SURVEY s.8d: one 256x256 image -> anomaly map


# Sources a committed PMC pass speaks for (profiles/r*_traffic.json carry the digest of these files as they were when the pass ran; bench.py
# attaches a pass to its line only while the digest still matches the tree -- VERDICT r5 item 6)
SCORE_TRAFFIC_SOURCES = ["self-supervised-anomaly-detection_amd/csrc/conv_igemm.hip", "self-supervised-anomaly-detection_amd/csrc/common.h"]
P16_TRAFFIC_SOURCES = ["self-supervised-anomaly-detection_amd/csrc/*.hip", "self-supervised-anomaly-detection_amd/csrc/common.h",
                       "self-supervised-anomaly-detection_amd/self_supervised/training.py"]


def source_sha(patterns):
    """sha256 over the named source files (repo-relative glob patterns, sorted paths, path + contents)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for pat in patterns:
        for f in sorted(glob.glob(os.path.join(ROOT, pat))):
            h.update(os.path.relpath(f, ROOT).encode() + b"\0")
            h.update(open(f, "rb").read())
    return h.hexdigest()


def pinned_traffic(pattern, match, sources):
    """The newest committed PMC pass matching `match` -> (profile dict or None, reason when dropped).  A pass whose recorded
    source digest differs from the tree's (or that carries none) is NOT attached: the kernel it measured is not this one."""
    import glob
    now = source_sha(sources)
    for tj in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        tjd = json.load(open(tj))
        if not match(tjd):
            continue
        rec = (tjd.get("source_sha") or {}).get("sha256")
        name = "profiles/" + os.path.basename(tj)
        if rec != now:
            return None, (f"{name} was taken on other kernel sources (recorded digest {str(rec)[:12]}, tree {now[:12]}): not attached; "
                          "re-run tools/profile_round.sh / tools/profile_p16.sh")
        return dict(tjd, file=name), None
    return None, "no committed PMC pass with this launch geometry"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step under weak scaling")
    ap.add_argument("--global-batch", type=int, default=256,
                    help="images per step over ALL ranks under strong scaling (SURVEY s.8e: 256)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="which partition is the headline when N > 1 (the other one is reported as an extra)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--phase", choices=["both", "train", "score"], default="both")
    ap.add_argument("--config", choices=["resnet18", "wrn50"], default="resnet18",
                    help="resnet18: BASELINE configs[1] / [2] (the headline); wrn50: configs[3], WideResNet-50-2 layer1-3 feature-distance "
                         "maps at 512x512 batch 64 on one GPU (throughput only: the reference has no such model)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end tools.training measurement (N = 1 only)")
    ap.add_argument("--no-wrn50", action="store_true", help="skip the BASELINE configs[3] (WideResNet-50 512x512 bs64) member of the N = 1 line")
    ap.add_argument("--no-partition-extra", action="store_true", help="skip the strong-partition / batch-32 extra line")
    ap.add_argument("--no-faithful", action="store_true", help="skip the bf16x6 (fp32-faithful split products) side measurement")
    ap.add_argument("--no-precision16", action="store_true", help="skip the precision16 member (the reference's default precision) of the N = 1 line")
    ap.add_argument("--no-graph", action="store_true", help="launch the training step eagerly instead of replaying hipGraphs")
    ap.add_argument("--train-precision", choices=["32", "16", "bf16"], default="32",
                    help="32: exact fp32 MFMA (headline); 16: fp16 operands + loss scaling (the reference's Trainer(precision=16)); "
                         "bf16: bf16 operands")
    ap.add_argument("--extras", default="", help="comma list of opt-in side measurements: precision16, bf16, bf16x3, bf16x6")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks as fresh child processes
    (this parent never touches the GPU and never re-execs itself) and return their exit status."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def build_model(dev, seed=0):
    import torch
    from self_supervised.models import PeraNet
    g = torch.Generator().manual_seed(seed)
    with torch.random.fork_rng():
        torch.manual_seed(seed)
        m = PeraNet()
    with torch.no_grad():        # random-init weights of the named architecture + non-trivial BN statistics
        for mod in m.modules():
            if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                mod.running_mean.copy_(0.1 * torch.randn(mod.num_features, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.num_features, generator=g))
    return m.to(dev)


def synth_images(n, size, seed, dev):
    import torch
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (n, 3, size, size), generator=g, dtype=torch.int32).float()
    k = torch.ones(3, 1, 3, 3) / 9.0
    u8 = torch.nn.functional.conv2d(torch.nn.functional.pad(u8, [1, 1, 1, 1], mode="replicate"), k, groups=3)
    x = u8.round().clamp(0, 255) / 255.0
    mean = torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)
    return ((x - mean) / std).contiguous().to(dev)


def score_batch(model, det, x, target=256):
    import torch
    from self_supervised import tools
    with torch.no_grad():
        emb = model(x)["latent_space"]
        det.batch = x.shape[0]
        maps = det.predict(emb)
        return tools.upsample(maps, target, verbose=False)


def host_cores():
    """Threads the CPU baseline may use: the cgroup CPU quota when one is set (the GPU box gives a 1-GPU job a
    16-CPU share of a 256-thread host), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return min(n, int(os.environ.get("SSAD_CPU_THREADS", "16")))


def _timed_median(fn, budget_s, warmups=2, max_iters=5, min_iters=1):
    """Median wall time of fn() over up to `max_iters` runs after `warmups` untimed ones, inside `budget_s` seconds:
    the iteration counts shrink (never below one warm-up + `min_iters` timed) when a single run is slow."""
    t0 = time.perf_counter()
    fn()
    first = time.perf_counter() - t0
    warm_done = 1
    while warm_done < warmups and (warm_done + 1 + min_iters) * first <= budget_s:
        fn(); warm_done += 1
    ts = []
    while len(ts) < max_iters and (len(ts) < min_iters or time.perf_counter() - t0 + first <= budget_s):
        t1 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t1)
    ts.sort()
    return ts[len(ts) // 2], warm_done, len(ts)


def cpu_baseline(args):
    """Oracle (torch-CPU fp32 restatement of the reference path, SURVEY s.8d) timed on the host cores at N = all
    granted cores and at 1 thread: a 256-image training step as 8 accumulated chunks of 32 (fwd + bwd per chunk, one SGD
    update), and U_map for 8 images; 2 warm-ups and the median of 5 where the time budget allows (the sample shrinks, and
    says so, where it does not)."""
    import torch
    from oracle import weights as ow, scoring as osc
    from oracle.peranet import OraclePeraNet, train_step, make_optimizer
    cores = host_cores()
    sd = ow.seeded_state_dict(0)
    bank = ow.synthetic_bank(588, 512, seed=2).numpy()
    out = {"unit": "images/sec", "kind": "port", "cores": cores}
    notes = []
    for threads, tag, n_train, n_img, budget in ((cores, "", 256, 8, 30.0), (1, "_1thread", 32, 1, 25.0)):
        torch.set_num_threads(threads)
        m = OraclePeraNet(); m.load_state_dict(sd)
        m.train()
        opt, _ = make_optimizer(m, 0.03, 10, "projection_train")
        xb, yb = ow.synthetic_images(32, args.size, seed=1234), ow.synthetic_labels(32, seed=1235)
        chunks = n_train // 32

        def train_once():
            opt.zero_grad()
            for _ in range(chunks):
                loss, _, _ = train_step(m, xb, yb)
                (loss / chunks).backward()
            opt.step()
        t_train, w, k = _timed_median(train_once, budget)
        notes.append(f"{threads} thread(s): train step of {n_train} images as {chunks} x 32 accumulated, {w} warm-up(s), median of {k}")
        m.eval(); m.patch_level = True
        xs = ow.synthetic_images(n_img, args.size, seed=4321)

        def score_once():
            with torch.no_grad():
                emb = m(xs)["latent_space"].numpy()
                s, _, _ = osc.cosine_knn_mean(bank, emb, 3)
                osc.upsample(torch.from_numpy(s).reshape(n_img, 1, 29, 29), args.size)
        t_score, w, k = _timed_median(score_once, budget)
        notes.append(f"scoring of {n_img} image(s) ({n_img * 841} patches, 588-row bank, blur + bilinear), {w} warm-up(s), median of {k}")
        out["value" + tag] = round(n_train / t_train, 2)
        out["anomaly_maps_per_sec" + tag] = round(n_img / t_score, 3)
    torch.set_num_threads(cores)
    out["sample"] = "oracle on torch-CPU fp32, " + str(args.size) + "x" + str(args.size) + "; " + "; ".join(notes)
    return out


def end_to_end(args):
    """tools.training with its DEFAULT arguments (the GPU input pipeline is the default since round 4) at the reference's own
    settings (tools.py:204-214: batch 96, image level, 256 x 256) on a synthetic MVTec-shaped category: images per second of whole
    fine-tune epochs -- host-side sampling of the defect parameters (8 sampler workers, as the reference's DataLoader has 8
    workers), GPU synthesis of the batch, training step, memory-bank gathering -- in fp32 and in the reference's
    Trainer(precision=16).  Median over the epochs after the first (which records the hipGraph).  Then tools.inference +
    tools.upsample on the 96 test PNGs of the same tree."""
    import contextlib
    import tempfile
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_mvtec import make_tree
    from self_supervised import tools
    res = {"workload": "tools.training(...) with default arguments (gpu_pipeline defaults to on), bottle-shaped synthetic category, "
                       "256x256 image level, batch 96, 1 + 5 epochs of ~10 steps; median fine-tune epoch after the first"}

    def step_only(prec):
        """The replayed training step alone at the SAME batch size (96, the reference's default, tools.py:208) with the batch resident:
        the like-for-like ceiling of the end-to-end figure (the headline step is batch 256 and 15-20 % faster per image)."""
        from self_supervised import training
        dev = torch.device("cuda", 0)
        m = build_model(dev)
        m.train(); m.unfreeze()
        st = training.DataParallelStep(m, lr=0.005, world_size=1, precision=prec)
        xs, ys = synth_images(96, args.size, 77, dev), torch.randint(0, 4, (96,), generator=torch.Generator().manual_seed(78)).to(dev)
        for _ in range(3):
            st.step(xs, ys)
        xb, yb = st.bind_inputs(xs, ys)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            st.step(xb, yb)
        torch.cuda.synchronize()
        return round(96 * 20 / (time.perf_counter() - t0), 1)
    with tempfile.TemporaryDirectory() as tmp:
        root = make_tree(os.path.join(tmp, "data"), categories=("bottle",), n_train=40, n_test_good=48, n_test_bad=48, size=256)
        for prec in (32, 16):
            with contextlib.redirect_stdout(sys.stderr):
                hist = tools.training(root + "bottle/", os.path.join(tmp, f"out{prec}") + "/", "bottle", imsize=(args.size, args.size),
                                      batch_size=96, seed=0, projection_training_params=(1, 0.03), fine_tune_params=(5, 0.005),
                                      trainer_kwargs={"precision": prec, "limit_val_batches": 1})
            rates = sorted(n / t for n, t in hist["throughput"]["fine_tune"][1:])
            res["fp32" if prec == 32 else "precision16"] = {
                "end_to_end_train_images_per_sec": round(rates[len(rates) // 2], 1),
                "step_only_batch96_images_per_sec": step_only(prec),
                "epochs": [[n, round(t, 4)] for n, t in hist["throughput"]["fine_tune"]],
                "projection_stage_images_per_sec": round(sum(n for n, _ in hist["throughput"]["projection_train"]) /
                                                         sum(t for _, t in hist["throughput"]["projection_train"]), 1)}
        # tools.inference as the reference calls it (tools.py:310-390): checkpoint -> PNG files -> predict -> bank from the first
        # training image -> k-NN maps -> blur + bilinear.  Checkpoint load, PNG decode and the copies back into the returned CPU
        # container are inside the clock: the figure is the rate a user of the API sees, beside the kernel-only rate.
        with contextlib.redirect_stdout(sys.stderr):
            ck = os.path.join(tmp, "out32") + "/best_model.ckpt"
            tools.inference(ck, root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)      # warm-up (plans, lazy init)
            torch.cuda.synchronize()
            import gc
            gc.collect()               # the training runs above left cycles behind; their frees would wait for the device inside the clock
            gc.disable()
            try:
                t0 = time.perf_counter()
                r = tools.inference(ck, root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)
                up = tools.upsample(r.anomaly_maps, args.size)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            finally:
                gc.enable()
        # tools.Evaluator on the upsampled maps: device-resident maps take the hand-written sort / scan kernels (csrc/auroc.hip), host
        # maps the reference's sklearn / numpy route -- both timed once, same scores
        def evaluate(maps):
            r.anomaly_maps = maps
            ev = tools.Evaluator(evaluation_metrics=["auroc", "aupro", "iou"])
            with contextlib.redirect_stdout(sys.stderr):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ev.evaluate(r, "bottle", None, patch_level=True)
                torch.cuda.synchronize()
            return time.perf_counter() - t1, [ev.scores.auroc, ev.scores.aupro, ev.scores.iou]
        evaluate(up)                                       # warm-up (lazy imports)
        t_dev, sc_dev = evaluate(up)
        t_host, sc_host = evaluate(up.cpu())
        res["evaluation"] = {"workload": "tools.Evaluator(['auroc', 'aupro', 'iou']).evaluate on the 96 upsampled 256x256 maps (6.3 M pixel scores)",
                             "seconds_device_maps": round(t_dev, 4), "seconds_host_maps": round(t_host, 3),
                             "max_abs_score_difference": float(max(abs(a - b) for a, b in zip(sc_dev, sc_host)))}
        res["inference"] = {"workload": "tools.inference(patch_localization=True) + tools.upsample on 96 test PNGs of 256x256 (+ the one training image that becomes the bank), "
                                        "default arguments; includes checkpoint load, PNG decode, the bank image and the returned CPU container",
                            "end_to_end_maps_per_sec": round(up.shape[0] / dt, 1), "images": int(up.shape[0]), "seconds": round(dt, 3)}
    return res


def measure_wrn50(args, steps=None, warmup=None):
    """BASELINE configs[3]: 64 synthetic 512 x 512 images -> WideResNet-50-2 layer1-3 features -> per-scale cosine 3-NN maps against
    588-row banks -> blur + bilinear 512 x 512, mean over the scales.  One GPU; throughput only (no reference counterpart).
    Returns the bench object (a full line under --config wrn50, the "wrn50" member of the default line otherwise)."""
    import torch
    from self_supervised import ops
    from self_supervised.wrn50 import FeatureDistanceScorer, WideResNet50Features
    from oracle import wrn50 as ow
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    dev = torch.device("cuda", 0)
    size, batch = 512, 64
    ref = ow.seeded_trunk(0)
    m = WideResNet50Features()
    m.load_state_dict(ref.state_dict())
    m.to(dev).eval()
    banks = ow.seeded_banks(588)
    scorer = FeatureDistanceScorer([b.to(dev) for b in banks])
    x = synth_images(batch, size, 1234, dev)

    def step():
        with torch.no_grad():
            return scorer(m(x), size)
    for _ in range(max(warmup, 1)):
        step()
    torch.cuda.synchronize()
    ops.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = ops.drain_profile()
    ops.PROFILE = None
    out = {"metric": "anomaly-maps/sec, WideResNet-50-2 layer1-3 feature-distance maps 512x512 bs64", "unit": "anomaly-maps/sec",
           "n_gpus": 1, "steps": steps, "warmup": warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "WideResNet-50 multi-scale (layer1-3) feature-distance maps, 512x512 bs64, 1 GPU (BASELINE configs[3]; "
                                  "no reference counterpart: the reference hard-wires resnet18, models.py:58-62)",
                      "definition": "torchvision wide_resnet50_2 (Bottleneck v1.5, width 128) up to layer3, eval-mode BatchNorm, random-init "
                                    "weights; per scale the reference's scorer (cosine 3-NN mean against a 588-row bank, models.py:345-370) on "
                                    "every pixel, then relu(gaussian_blur k=7) + bilinear to 512 (tools.py:394-399); mean over the three scales "
                                    "(self_supervised/wrn50.py, oracle/wrn50.py)",
                      "images_per_gpu": batch, "bank_rows": 588, "scales": [[128, 256], [64, 512], [32, 1024]]},
           "value": round(batch * steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 3)}
    recs = [r for r in prof if r["kernel"].startswith("conv_igemm") or r["kernel"].startswith("conv3x3_fw32")]
    t = sum(r["ms"] for r in recs) * 1e-3
    fl = sum(r["flops"] for r in recs)
    allk = sum(r["ms"] for r in prof) * 1e-3
    out["roofline"] = {"bound": "mfma", "achieved": round(fl / t / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(fl / t / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                       "kernel": "the trunk's convs: conv_igemm_f32_kernel (NHWC: 1 x 1 and stride-2 convs) + conv3x3_hw_kernel<float> (3 x 3 / "
                                 "stride 1, csrc/conv16w.hip); every FLOP counted is issued, no tap skipping",
                       "launches": len(recs), "avg_launch_ms": round(1e3 * t / len(recs), 4),
                       "alg_gflop_per_launch": round(fl / len(recs) / 1e9, 3), "share_of_gpu_time": round(t / allk, 4),
                       "gflop_per_image": round(sum(r["flops"] for r in prof) / steps / batch / 1e9, 2)}
    by = {}
    for r in prof:
        e = by.setdefault(r["kernel"], [0.0, 0, 0.0]); e[0] += r["ms"]; e[1] += 1; e[2] += r["flops"]
    out["kernel_ms"] = {k: [round(v[0] / steps, 3), v[1] // steps, round(v[2] / max(v[0], 1e-9) / 1e9, 1)] for k, v in sorted(by.items())}
    if not args.no_cpu_baseline:
        cores = host_cores()
        torch.set_num_threads(cores)
        xc = x[:1].cpu()

        def once():
            with torch.no_grad():
                ow.distance_maps(ref(xc), banks, size)
        tt, w, k = _timed_median(once, 15.0, max_iters=3)
        out["cpu_baseline"] = {"value": round(1.0 / tt, 3), "unit": "anomaly-maps/sec", "cores": cores, "kind": "port",
                               "sample": f"oracle/wrn50.py on torch-CPU fp32: 1 image 512x512 (trunk + three k-NN maps + blur / bilinear), "
                                         f"{w} warm-up(s), median of {k}"}
    del m, scorer, x
    torch.cuda.empty_cache()
    return out


def bench_wrn50(args):
    print(json.dumps(measure_wrn50(args)))


def main():
    args = parse_args()
    if args.config == "wrn50":
        return bench_wrn50(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    # ranks started by an external torchrun get the IPC mode launch_ranks() sets (the host driver only supports dmabuf IPC: without it
    # RCCL fails with hipIpcGetMemHandle: invalid argument) -- before torch is imported
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # one rank per GPU; SSAD_DIST_BACKEND=gloo + fewer GPUs than ranks is only for rehearsing the N>1 code path on a
    # one-GPU box (ranks then share device 0 and gloo stages the all-reduce through the host)
    backend = os.environ.get("SSAD_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from self_supervised import ops, training
    from self_supervised.models import AnomalyDetector

    def rccl_probe():
        """First contact of a multi-GPU run, OUTSIDE every timed region: which (rank, device) pairs are in the job (an all-gather of
        rank, local device index, device uuid), the collective library and its version, and the bus bandwidth of ONE all-reduce of the
        step's gradient payload (12 691 524 fp32 = 50.8 MB; busbw = 2 (n - 1) / n x bytes / time, the ring figure of merit).  A run
        that later stalls or scales badly then says what it ran on."""
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device": dev.index, "uuid": str(getattr(props, "uuid", "")), "name": props.name}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        n_el = 12691524
        buf = torch.ones(n_el, device=dev, dtype=torch.float32)
        for _ in range(2):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        dist.barrier()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        ver = None
        if backend == "nccl":
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:          # noqa: BLE001
                ver = f"unknown ({type(e).__name__})"
        uu = [s["uuid"] for s in seen]
        return {"backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, "version": ver, "ranks_seen": seen,
                "distinct_devices": len(set((s["uuid"] or s["device"]) for s in seen)) if all(uu) else len(set(s["device"] for s in seen)),
                "allreduce_probe": {"bytes": 4 * n_el, "seconds": round(dt, 6), "algbw_GBps": round(4 * n_el / dt / 1e9, 2),
                                    "busbw_GBps": round(2 * (world - 1) / world * 4 * n_el / dt / 1e9, 2),
                                    "note": "one fp32 all-reduce of the gradient payload, mean of 5 after 2 warm-ups, max over ranks; "
                                            "xGMI: 7 links x ~153 GB/s per GPU, a ring is bound by one link per direction"}}

    strong_headline = args.scaling == "strong"
    if strong_headline:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} does not divide over {world} ranks")
        per_rank = args.global_batch // world
    else:
        per_rank = args.batch
    os.environ.setdefault("SSAD_ALLOW_RANDOM_BACKBONE", "1")      # synthetic benchmark: random-init weights by design
    model = build_model(dev)
    x = synth_images(per_rank, args.size, 1234 + rank, dev)
    y = torch.randint(0, 4, (per_rank,), generator=torch.Generator().manual_seed(1235 + rank)).to(dev)
    bank = torch.randn(588, 512, generator=torch.Generator().manual_seed(2)).to(dev)
    use_graph = not args.no_graph

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        import gc
        for _ in range(warmup):
            fn()
        # as timeit does: no collector pass (whose frees wait for the device) inside the timed steps.  (A gc.collect() right here was
        # measured too: the batch-32 step that follows it runs 2 % slower, 5.58-5.62 against 5.47 ms -- the collector is only switched off.)
        gc.disable()
        try:
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            barrier()
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    def optional(name, fn):
        """Side measurements must never take the headline line down with them: on failure they are reported as skipped
        (all ranks run the same code, so a deterministic failure is collective and nobody is left waiting)."""
        try:
            return fn()
        except Exception as e:          # noqa: BLE001
            print(f"[bench] optional measurement '{name}' skipped: {type(e).__name__}: {e}", file=sys.stderr)
            return None

    res, prof = {}, {}
    if world > 1:
        res["rccl"] = optional("rccl_probe", rccl_probe)
    extras = [e for e in args.extras.split(",") if e]
    # bf16x6 (every fp32 product formed from three bf16 parts per operand, the six largest partial products, fp32 accumulate: the
    # accuracy of the fp32 MFMA at 6/16 of its matrix time) is measured by default on one GPU beside the exact-fp32 headline
    faithful = world == 1 and not args.no_faithful and args.train_precision == "32"
    if faithful and "bf16x6" not in extras:
        extras.append("bf16x6")
    if args.phase in ("both", "train"):
        model.train()
        model.unfreeze()
        prec = {"32": 32, "16": 16, "bf16": "bf16"}[args.train_precision]
        trainer = training.DataParallelStep(model, lr=0.005, world_size=world, precision=prec, graph=use_graph)
        ops.PROFILE = None
        # start-up self-check (every N; decisive at N > 1): one eager step and one graph-replayed step from the same state must
        # leave identical bits, and every rank the same replica -- else all ranks fall back to eager launches and the line says so
        res["self_check"] = optional("self_check", lambda: trainer.self_check(x, y))
        res["launch_mode"] = trainer.launch_mode
        # warm up (eager step, capture), then let the batch live in the recorded step's own input buffer: a producer (the GPU input
        # pipeline) writes there directly, so the timed steps carry no device-to-device copy of the batch
        for _ in range(max(args.warmup, 2)):
            trainer.step(x, y)
        xb, yb = trainer.bind_inputs(x, y)
        res["train_s"] = timed(lambda: trainer.step(xb, yb), args.steps, 1)
        res["train_graph_segments"] = sum(1 for p in trainer._plans.values() for o in p["ops"] if o[0] == "graph")
        if world > 1:
            # how much of the gradient all-reduce stays exposed: HIP events around the wait of every step (a few steps outside the
            # timed region: event records are not free), and the bytes each rank puts on the wire per step
            def comm_probe():
                trainer.comm_events = []
                for _ in range(max(3, min(args.steps, 10))):
                    trainer.step(x, y)
                torch.cuda.synchronize()
                ms = [a.elapsed_time(b) for a, b in trainer.comm_events]
                trainer.comm_events = None
                t = torch.tensor([sum(ms) / max(len(ms), 1)], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return {"comm_exposed_ms": round(t.item(), 4), "allreduce_bytes": int(trainer.allreduce_bytes()),
                        "allreduce_messages": int(trainer.allreduce_messages())}
            res["comm"] = optional("comm_probe", comm_probe)
        # per-kernel attribution: the same step launched eagerly with HIP events around every launch (a captured graph
        # cannot carry them); not part of the timed region above
        ops.PROFILE = []
        for _ in range(2):
            trainer.step(x, y)
        prof["train"] = ops.drain_profile()
        prof_train_steps = 2
        ops.PROFILE = None

        def partition_extra():
            nb = 32 if world == 1 else (args.batch if strong_headline else args.global_batch // world)
            if nb == per_rank or nb < 1:
                return None
            xs, ys = x[:nb].contiguous(), y[:nb].contiguous()
            if xs.shape[0] < nb:
                xs = synth_images(nb, args.size, 99 + rank, dev)
                ys = torch.randint(0, 4, (nb,), generator=torch.Generator().manual_seed(98 + rank)).to(dev)
            for _ in range(3):
                trainer.step(xs, ys)
            xsb, ysb = trainer.bind_inputs(xs, ys)
            dt = timed(lambda: trainer.step(xsb, ysb), max(args.steps, 10), 1)
            return {"images_per_gpu": nb, "global_batch": nb * world, "steps": max(args.steps, 10),
                    "train_images_per_sec": round(world * nb * max(args.steps, 10) / dt, 2),
                    "train_ms_per_step": round(1e3 * dt / max(args.steps, 10), 3)}
        if not args.no_partition_extra:
            r = optional("partition", partition_extra)
            if r:
                res["partition"] = r
        def precision16_object():
            """The reference's DEFAULT training precision (pl.Trainer(precision=16), tools.py:263, :296) as a first-class figure: the
            step with fp16 operands, fp32 accumulation, dynamic loss scaling and -- since round 5 -- every trunk activation and its
            gradient stored as halves, as torch.autocast stores them.  Its matrix work is a few per cent of the step (16-bit MFMAs);
            what bounds it is bytes: `roofline` prices SURVEY 8d's algorithmic bytes (2-byte activations, fp32 parameters / optimizer)
            against HBM, `mfma_frac` the algorithmic FLOPs against the dense fp16 MFMA peak."""
            tx = training.DataParallelStep(model, lr=0.005, world_size=world, precision=16, graph=use_graph)
            n = max(args.steps, 10)
            for _ in range(3):                                    # eager step, capture, one replay
                tx.step(x, y)
            xb16, yb16 = tx.bind_inputs(x, y)
            dt = timed(lambda: tx.step(xb16, yb16), n, 1)
            ms = 1e3 * dt / n
            ops.PROFILE = []
            for _ in range(2):
                tx.step(xb16, yb16)
            pr = ops.drain_profile()
            ops.PROFILE = None
            by = {}
            for r in pr:
                e = by.setdefault(r["kernel"], [0.0, 0, 0.0, 0.0]); e[0] += r["ms"]; e[1] += 1; e[2] += r["flops"]; e[3] += r["bytes"]
            # PMC passes cannot run inside this process: the newest committed pass with this batch size, IF it was taken on these sources
            tjd, why16 = pinned_traffic("r*_p16_traffic.json", lambda d: d.get("images_per_step") == per_rank, P16_TRAFFIC_SOURCES)
            tprof = None
            if tjd:
                tprof = {"file": tjd["file"], "traffic_MB_per_step": tjd["traffic_MB_per_step"],
                         "ratio_to_algorithmic": tjd.get("ratio_to_algorithmic"), "counters": tjd.get("correction"),
                         "source_sha256": tjd["source_sha"]["sha256"]}
            act_mb = per_rank * U_TRAIN_ACT_MB_F32 / 2
            alg_gb = (act_mb + OPTIMIZER_MB_PER_STEP) / 1e3
            kern_gb = sum(r["bytes"] for r in pr) / 2 / 1e9
            out16 = {
                "arithmetic": "fp16 operands / fp32 accumulate (v_mfma_f32_32x32x16_f16), activations and their gradients stored as halves, "
                              "fp32 master weights, statistics, loss and SGD, device-side GradScaler",
                "half_tensors": bool(tx.eng.h16), "images_per_gpu": per_rank,
                "train_images_per_sec": round(world * per_rank * n / dt, 2), "train_ms_per_step": round(ms, 3), "steps": n,
                "roofline": {"bound": "hbm", "achieved": round(alg_gb / (ms * 1e-3), 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                             "frac": round(alg_gb / (ms * 1e-3) / PEAK_HBM_GBPS, 4),
                             "traffic": round(tprof["traffic_MB_per_step"] * 1e6) if tprof else None,
                             "traffic_unit": "fabric-side bytes per STEP, all kernels (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 PMC passes; "
                                             "Infinity-Cache hits included)" if tprof else None,
                             "traffic_profile": tprof if tprof else {"dropped": why16},
                             "alg_GB_per_step": round(alg_gb, 3),
                             "definition": f"SURVEY 8d: {per_rank} images x 3 x 33.55 MB / 2 (half activations: forward, input gradients, "
                                           f"weight-gradient reads) + {OPTIMIZER_MB_PER_STEP} MB optimizer traffic, over the replayed step's wall time",
                             "kernel_sum_GB_per_step": round(kern_gb, 3),
                             "kernel_sum_note": "sum over the step's launches of each kernel's own operand bytes (in + out once): what this "
                                                "kernel decomposition moves at least, un-fused BatchNorm passes included"},
                "mfma_frac": round(world * per_rank * n / dt / world * U_TRAIN_GFLOP / 1e3 / PEAK_F16_MFMA_TFLOPS, 4),
                "mfma_peak_TFLOPs": PEAK_F16_MFMA_TFLOPS,
                "kernel_ms": {k: [round(v[0] / 2, 3), v[1] // 2, round(v[2] / max(v[0], 1e-9) / 1e9, 1), round(v[3] / max(v[0], 1e-9) / 1e6, 1)]
                              for k, v in sorted(by.items())},
                "kernel_ms_columns": "ms per step, launches per step, TFLOP/s, GB/s (algorithmic bytes of the kernel's operands)"}
            del tx
            return out16
        # (N = 1 only by default: a second DataParallelStep brings a second capture stream, and with two of them a multi-rank replay
        # was measured to stall on this runtime -- DESIGN s.6; `--extras precision16` forces it)
        if args.train_precision == "32" and not args.no_precision16 and (world == 1 or "precision16" in extras):
            extras = [e for e in extras if e != "precision16"]
            r16 = optional("precision16", precision16_object)
            if r16:
                res["precision16"] = r16
            trainer.eng.bf16 = training.precision_mode(prec)
        for name in extras:
            pmap = {"precision16": 16, "bf16": "bf16", "bf16x3": "bf16x3", "bf16x6": "bf16x6"}
            if name not in pmap:
                continue

            def run_extra(p=pmap[name]):
                tx = training.DataParallelStep(model, lr=0.005, world_size=world, precision=p, graph=use_graph)
                return timed(lambda: tx.step(x, y), max(args.steps, 10), 4)       # eager step, capture, two replays, then the clock
            dt = optional(name, run_extra)
            if dt:
                res["train_extra_" + name] = round(world * per_rank * max(args.steps, 10) / dt, 2)
            trainer.eng.bf16 = training.precision_mode(prec)
        del trainer
    # the training step objects hold reference cycles (closures over the engine): collect them NOW, so that their hipGraphs and
    # private memory pools are released here and not by a garbage-collector pass in the middle of the timed scoring passes
    # (a hipFree waits for the device: measured 387 ms per pass instead of 375 in the first process on a fresh box)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    if args.phase in ("both", "score"):
        model.eval(); model.enable_patch_level_mode()
        det = AnomalyDetector(patch_level=True, batch=per_rank, num_patches=841)
        det.fit_bank(bank)
        ops.PROFILE = None
        for i in range(max(args.warmup, 1)):
            if i == max(args.warmup, 1) - 1:
                # the last warm-up pass runs exactly what the timed passes run, HIP events around every launch included: in the first
                # process on a fresh box the event path is otherwise executed -- and paged in from the image -- for the first time
                # inside the timed region (measured: 387.8 ms per pass in the first process of a box, 375.4 in every later one)
                ops.PROFILE = []
            score_batch(model, det, x, args.size)
        ops.drain_profile()
        ops.PROFILE = []           # scoring is ~270 launches of ~1.5 ms: the events ride inside the timed region
        res["score_s"] = timed(lambda: score_batch(model, det, x, args.size), args.steps, 0)
        prof["score"] = ops.drain_profile()
        ops.PROFILE = None
        if world == 1 and per_rank > 32 and not args.no_partition_extra:
            # scoring side of the 8-GPU strong partition: 32 images = 26 912 patches per rank
            def score32():
                xs = x[:32].contiguous()
                d32 = AnomalyDetector(patch_level=True, batch=32, num_patches=841)
                d32.fit_bank(bank)
                n = max(args.steps, 5)
                dt = timed(lambda: score_batch(model, d32, xs, args.size), n, 1)
                return {"anomaly_maps_per_sec": round(32 * n / dt, 2), "score_ms_per_step": round(1e3 * dt / n, 3), "steps": n}
            res["score32"] = optional("batch-32 scoring", score32)
        if "bf16x6" in extras:                 # how far the bf16x6 maps are from the exact-fp32 ones on this very batch
            def map_diff():
                ref = score_batch(model, det, x[:32].contiguous(), args.size)
                os.environ["SSAD_MATH"] = "bf16x6"
                try:
                    got = score_batch(model, det, x[:32].contiguous(), args.size)
                finally:
                    os.environ["SSAD_MATH"] = "f32"
                return [float((got - ref).abs().max()), float(ref.abs().max())]
            res["x6_map_diff"] = optional("bf16x6 map difference", map_diff)
        for tag in ("bf16x3", "bf16x6"):
            if tag in extras:
                os.environ["SSAD_MATH"] = tag

                def run_score_extra():
                    return timed(lambda: score_batch(model, det, x, args.size), args.steps, 1)
                dt = optional("score_" + tag, run_score_extra)
                if dt:
                    res["score_extra_" + tag] = round(world * per_rank * args.steps / dt, 3)
                os.environ["SSAD_MATH"] = "f32"
        model.disable_patch_level_mode()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    mode = "strong" if strong_headline else "weak"
    rehearsal = world > 1 and (backend != "nccl" or ndev < world)
    hw = f"{world}xMI355X" if not rehearsal else f"{world} ranks sharing {max(ndev, 1)} GPU(s) over {backend} (REHEARSAL of the N > 1 code path, not a scaling figure)"
    out = {
        "metric": "train images/sec + anomaly-maps/sec, ResNet-18 256x256 bs256",
        "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "higher_is_better": True, "scaling": mode, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"ResNet-18 {args.size}x{args.size} bs{per_rank * world if strong_headline else per_rank} self-sup train + "
                               f"anomaly map, {hw}, synthetic images (BASELINE configs[{1 if world == 1 else 2}])",
                   "images_per_gpu": per_rank, "global_batch": per_rank * world, "patches_per_image": 841, "bank_rows": 588,
                   "parallelism": f"dp{world} ({mode}: {per_rank} images per rank, global batch {per_rank * world}; bucketed gradient "
                                  f"all-reduce overlapped with backward)" if world > 1 else f"dp1 ({per_rank} images)",
                   "train_step_launch": res.get("launch_mode", "hipGraph segments" if use_graph else "eager"),
                   "input_binding": "in place: the timed steps read the batch from the recorded step's own input buffer (DataParallelStep.bind_inputs; "
                                    "no device-to-device copy of the batch inside the clock, warm-up outside it).  Rounds 1-4 timed step(x, y) "
                                    "with a 201 MB copy per 256-image step (~0.05 ms): their img/s are lower by that much and otherwise comparable"},
    }
    if args.train_precision != "32":
        out["dtype"] = {"16": "f16 operands / f32 accumulate", "bf16": "bf16 operands / f32 accumulate"}[args.train_precision]
    tot_s = 0.0
    if "train_s" in res:
        out["value"] = round(world * per_rank * args.steps / res["train_s"], 2)
        out["train_ms_per_step"] = round(1e3 * res["train_s"] / args.steps, 3)
        out["train_frac_of_f32_mfma_peak"] = round(out["value"] / world * U_TRAIN_GFLOP / 1e3 / PEAK_F32_MFMA_TFLOPS, 4)
        out["config"]["train_graph_segments"] = res["train_graph_segments"]
        if res.get("self_check"):
            out["self_check"] = res["self_check"]
        if res.get("rccl"):
            out["rccl"] = res["rccl"]
        if res.get("comm"):
            out["comm"] = dict(res["comm"], note="comm_exposed_ms: HIP events around the wait for the bucketed all-reduce, max over ranks, "
                                                 "mean over steps; allreduce_bytes: fp32 gradient bytes per rank per step")
        if res.get("precision16"):
            out["precision16"] = res["precision16"]
        tot_s += res["train_s"]
    if "partition" in res:
        out["batch32" if world == 1 else ("weak" if strong_headline else "strong")] = res["partition"]
    if res.get("score32"):
        b32 = out.setdefault("batch32", {"images_per_gpu": 32, "global_batch": 32})
        b32["anomaly_maps_per_sec"] = res["score32"]["anomaly_maps_per_sec"]
        b32["score_ms_per_step"] = res["score32"]["score_ms_per_step"]
    if faithful and ("train_extra_bf16x6" in res or "score_extra_bf16x6" in res):
        f = {"arithmetic": "every fp32 product as the six largest partial products of three bf16 parts per operand "
                           "(v_mfma_f32_32x32x16_bf16), fp32 accumulate; weight gradients and everything outside the matrix cores exact fp32; "
                           "what is dropped is ~2^-25 of a product (tests/test_hip_x3.py holds it to the bars of the exact path); "
                           "SSAD_MATH=bf16x6 / Trainer(precision='bf16x6') select it, the default and the headline stay exact fp32"}
        if "train_extra_bf16x6" in res:
            f["train_images_per_sec"] = res["train_extra_bf16x6"]
            f["train_ms_per_step"] = round(1e3 * per_rank / res["train_extra_bf16x6"], 3)
        if "score_extra_bf16x6" in res:
            f["anomaly_maps_per_sec"] = res["score_extra_bf16x6"]
        if res.get("x6_map_diff"):
            f["max_abs_map_difference_to_f32"] = res["x6_map_diff"][0]
            f["max_abs_map_value"] = res["x6_map_diff"][1]
        out["bf16x6"] = f
    for k, v in res.items():
        if k.startswith("train_extra_") or k.startswith("score_extra_"):
            out.setdefault("extras", {})[k] = v
    if "score_s" in res:
        out["anomaly_maps_per_sec"] = round(world * per_rank * args.steps / res["score_s"], 3)
        out["score_ms_per_step"] = round(1e3 * res["score_s"] / args.steps, 3)
        tot_s += res["score_s"]
        if "value" not in out:
            out["value"], out["unit"] = out["anomaly_maps_per_sec"], "anomaly-maps/sec"
            out["metric"] = "anomaly-maps/sec, ResNet-18 256x256 bs256 (scoring phase only)"
    out["ms_per_step"] = round(1e3 * tot_s / args.steps, 3)

    # roofline of the dominant kernel of each phase, live HIP events on the launch stream.  Scoring: the position-major
    # instantiations of the implicit-GEMM conv (layers 2-4, 30 launches per pass); training: its NHWC instantiations (forward +
    # input-gradient convs of layers 2-4, linear layers).  The top-level "roofline" is the scoring one when scoring ran.
    def latest_traffic(samples):
        """HBM-side bytes per launch of the position-major kernel need rocprofv3 PMC passes, which cannot run inside this
        process: the newest committed, counter-corrected pass of this command line with the same launch geometry."""
        tjd, why = pinned_traffic("r[0-9][0-9]_traffic.json", lambda d: d.get("patches_per_launch") == samples, SCORE_TRAFFIC_SOURCES)
        if tjd is None:
            return {"dropped": why}
        return {"file": tjd["file"], "traffic_MB_per_launch": tjd["traffic_MB_per_launch"],
                "ratio_to_algorithmic": tjd.get("ratio_to_algorithmic"), "counters": tjd.get("correction"),
                "source_sha256": tjd["source_sha"]["sha256"]}

    def roofline_of(phase, tag):
        recs = [r for r in prof.get(phase, []) if r["kernel"] == tag]
        if not recs:
            return None
        t = sum(r["ms"] for r in recs) * 1e-3
        fl = sum(r["flops"] for r in recs)
        xfl = sum(r["exec_flops"] for r in recs)
        allk = sum(r["ms"] for r in prof[phase]) * 1e-3
        tiles = {}
        for r in recs:
            e = tiles.setdefault(r.get("tile") or "<small-batch linear / other>", [0, 0.0, 0.0]); e[0] += 1; e[1] += r["ms"]; e[2] += r["exec_flops"]
        inst = {k: {"launches": v[0], "ms": round(v[1], 3), "TFLOPs": round(v[2] / max(v[1], 1e-9) / 1e9, 1)}
                for k, v in sorted(tiles.items(), key=lambda kv: -kv[1][1])}
        # `frac` prices the MFMA FLOPs the kernel really ISSUES against the dense fp32-MFMA peak: position-major convs skip the
        # filter taps that fall into the zero padding (exact: only x * 0 products are dropped), so the algorithmic count of
        # SURVEY s.8d (2 M K Cout per conv) over the kernel's time can exceed the peak; that figure is kept as alg_frac.
        tprof = latest_traffic(getattr(model, "last_pass_samples", None)) if phase == "score" else None
        rf = {"bound": "mfma", "achieved": round(xfl / t / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
              "frac": round(xfl / t / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
              "alg_achieved": round(fl / t / 1e12, 2), "alg_frac": round(fl / t / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
              "skipped_tap_share": round(1.0 - xfl / fl, 4),
              "traffic": round(tprof["traffic_MB_per_launch"] * 1e6) if tprof and "dropped" not in tprof else None,
              "traffic_unit": "fabric-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, rocprofv3 PMC passes of this command; "
                              "Infinity-Cache hits included)" if tprof and "dropped" not in tprof else None,
              "traffic_profile": tprof,
              "kernel": ("conv3x3_hw_kernel<float> (csrc/conv16w.hip) x %d" % len(recs)) if tag == "conv3x3_fw32" else
                        "conv_igemm_f32_kernel" + " + ".join(f"{k} x {v['launches']}" for k, v in inst.items()),
              "instantiations": inst, "phase": phase, "launches": len(recs),
              "avg_launch_ms": round(1e3 * t / len(recs), 4),
              "alg_gflop_per_launch": round(fl / len(recs) / 1e9, 3),
              "exec_gflop_per_launch": round(xfl / len(recs) / 1e9, 3),
              "alg_MB_per_launch": round(sum(r["bytes"] for r in recs) / len(recs) / 1e6, 1),
              "alg_GBps": round(sum(r["bytes"] for r in recs) / t / 1e9, 1),
              "share_of_gpu_time": round(t / allk, 4)}
        return rf
    # the training pass's dominant matrix kernel: the register-fed 3x3 conv (csrc/conv16w.hip, T = float) where it runs, else the implicit GEMM
    tms = {}
    for r in prof.get("train", []):
        if r["kernel"] in ("conv3x3_fw32", "conv_igemm_f32"):
            tms[r["kernel"]] = tms.get(r["kernel"], 0.0) + r["ms"]
    train_tag = max(tms, key=tms.get) if tms else "conv_igemm_f32"
    rf_score, rf_train = roofline_of("score", "conv_igemm_pos_f32"), roofline_of("train", train_tag)
    if rf_score:
        # the same kernel template runs two kinds of launches since round 6: the convs of layers 2-4 (every position of 8 x 8 .. 2 x 2
        # maps: the launches of rounds 1-5) and the RING launches of layer1 (border positions of 16 x 16 maps, 64 channels, 8-18 K-steps
        # per workgroup).  `frac` covers all of them; the split is given so that rounds stay comparable
        ring = [r for r in prof.get("score", []) if r["kernel"] == "conv_igemm_pos_f32" and (r.get("tile") or "").startswith("<128,64,")]
        rest = [r for r in prof.get("score", []) if r["kernel"] == "conv_igemm_pos_f32" and r not in ring]
        for name, recs in (("layers2_4", rest), ("layer1_ring", ring)):
            if recs:
                t = sum(r["ms"] for r in recs) * 1e-3
                rf_score[name] = {"launches": len(recs), "ms_per_step": round(1e3 * t / args.steps, 3),
                                  "TFLOPs": round(sum(r["exec_flops"] for r in recs) / t / 1e12, 2),
                                  "frac": round(sum(r["exec_flops"] for r in recs) / t / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
    if rf_score or rf_train:
        out["roofline"] = dict(rf_score or rf_train)
        out["roofline"]["note"] = ("achieved / frac = MFMA FLOPs issued per second of kernel time (HIP events on the launch stream) "
                                   "over the dense fp32-MFMA peak; alg_* count SURVEY 8d's 2 M K Cout, which includes the taps in "
                                   "the zero padding that the position-major kernel skips (skipped_tap_share).")
        if rf_score and rf_train:
            out["roofline_train"] = rf_train
    # whole-pass fractions: the MFMA FLOPs every kernel of a pass issues over the pass's wall time (not only the dominant kernel's)
    wp = {}
    if "train" in prof and "train_s" in res:
        xf = sum(r["exec_flops"] for r in prof["train"]) / prof_train_steps
        wp["train"] = round(xf / (res["train_s"] / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
    if "score" in prof and "score_s" in res:
        xf = sum(r["exec_flops"] for r in prof["score"]) / args.steps
        wp["score"] = round(xf / (res["score_s"] / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
    if "score_s" in res:
        # the same pass by SURVEY 8d's ALGORITHMIC count (252.06 GFLOP per map: every tap of every patch), whatever was executed: padding
        # taps are skipped and, since round 6, layer1 arithmetic shared by overlapping patches is done once per image -- this can exceed 1
        wp["score_alg"] = round(out["anomaly_maps_per_sec"] / world * U_MAP_GFLOP / 1e3 / PEAK_F32_MFMA_TFLOPS, 4)
    if wp:
        out["whole_pass_frac"] = dict(wp, note="train / score: executed MFMA FLOPs of ALL kernels of the pass / its wall time / the dense fp32-MFMA "
                                               "peak; score_alg: SURVEY 8d's algorithmic FLOPs per map x maps/s / the same peak")
    if prof:
        # HBM-bound kernels: algorithmic bytes (each operand read once, each result written once) over their event time
        hbm = {}
        for ph in prof:
            for r in prof[ph]:
                if r["flops"] == 0.0 and r["bytes"] > 0:
                    e = hbm.setdefault(ph + "." + r["kernel"], [0.0, 0.0, 0]); e[0] += r["bytes"]; e[1] += r["ms"]; e[2] += 1
        out["hbm_kernels"] = {k: {"GBps": round(v[0] / max(v[1], 1e-9) / 1e6, 1), "frac_of_8TBps": round(v[0] / max(v[1], 1e-9) / 1e6 / PEAK_HBM_GBPS, 4),
                                  "launches": v[2], "MB_per_launch": round(v[0] / v[2] / 1e6, 2)} for k, v in sorted(hbm.items())}
        out["kernel_ms"] = {}
        for ph in prof:
            nst = prof_train_steps if ph == "train" else args.steps
            by = {}
            for r in prof[ph]:
                e = by.setdefault(r["kernel"], [0.0, 0, 0.0]); e[0] += r["ms"]; e[1] += 1; e[2] += r["flops"]
            out["kernel_ms"][ph] = {k: [round(v[0] / nst, 3), v[1] // nst,
                                        round(v[2] / max(v[0], 1e-9) / 1e9, 1)] for k, v in sorted(by.items())}
    if not args.no_e2e and world == 1 and args.phase in ("both", "train"):
        r = optional("end_to_end", lambda: end_to_end(args))
        if r:
            out["end_to_end"] = r
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args)
    if world == 1 and not args.no_wrn50 and args.phase == "both":
        model = det = trainer = None
        torch.cuda.empty_cache()
        w50 = optional("wrn50", lambda: measure_wrn50(args, steps=max(min(args.steps, 10), 3), warmup=2))
        if w50:
            for k in ("n_gpus", "higher_is_better", "scaling", "vs_baseline", "data"):
                w50.pop(k, None)
            out["wrn50"] = w50
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
