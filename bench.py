#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): train images/sec + anomaly-maps/sec, ResNet-18, 256x256, batch 256.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of 256 synthetic 256x256 images per GPU, inputs already
resident in HBM: (a) one pretext training step (forward + backward + SGD update, + RCCL gradient all-reduce when
N > 1) and (b) anomaly-map scoring of the same batch (841 sliding-window patches per image -> trunk -> 512-d
embedding -> cosine 3-NN against a 588-row bank -> blur -> bilinear 256x256 map).  Both rates are reported;
``value`` is the training rate (the first-named metric), ``anomaly_maps_per_sec`` the scoring rate.
Weak scaling: every rank processes its own 256-image batch.  fp32 throughout (exact f32 MFMA).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "self-supervised-anomaly-detection_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBPS = 8000.0


def build_model(dev, seed=0):
    from self_supervised.models import PeraNet
    g = torch.Generator().manual_seed(seed)
    m = PeraNet()
    with torch.no_grad():        # random-init weights of the named architecture + non-trivial BN statistics
        for mod in m.modules():
            if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                mod.running_mean.copy_(0.1 * torch.randn(mod.num_features, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.num_features, generator=g))
    return m.to(dev)


def synth_images(n, size, seed, dev):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (n, 3, size, size), generator=g, dtype=torch.int32).float()
    k = torch.ones(3, 1, 3, 3) / 9.0
    u8 = torch.nn.functional.conv2d(torch.nn.functional.pad(u8, [1, 1, 1, 1], mode="replicate"), k, groups=3)
    x = u8.round().clamp(0, 255) / 255.0
    mean = torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)
    return ((x - mean) / std).contiguous().to(dev)


def score_batch(model, det, x, target=256):
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


def cpu_baseline(args):
    """Oracle (torch-CPU fp32 restatement of the reference path) timed on the host cores: bounded sample."""
    from oracle import weights as ow, scoring as osc
    from oracle.peranet import OraclePeraNet, train_step, make_optimizer
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = ow.seeded_state_dict(0)
    m = OraclePeraNet(); m.load_state_dict(sd)
    # training: batch 16 @ 256^2, fwd + bwd + SGD step
    m.train()
    opt, _ = make_optimizer(m, 0.03, 10, "projection_train")
    xb, yb = ow.synthetic_images(16, 256, seed=1234), ow.synthetic_labels(16, seed=1235)
    for it in range(3):
        if it == 1:
            t0 = time.perf_counter()
        opt.zero_grad(); loss, _, _ = train_step(m, xb, yb); loss.backward(); opt.step()
    t_train = (time.perf_counter() - t0) / 2
    # scoring: 2 images -> 1682 patches -> kNN(588) -> blur -> bilinear
    m.eval(); m.patch_level = True
    bank = ow.synthetic_bank(588, 512, seed=2).numpy()
    xs = ow.synthetic_images(2, 256, seed=4321)
    with torch.no_grad():
        for it in range(2):
            t0 = time.perf_counter()
            emb = m(xs)["latent_space"].numpy()
            s, _, _ = osc.cosine_knn_mean(bank, emb, 3)
            osc.upsample(torch.from_numpy(s).reshape(2, 1, 29, 29), 256)
            t_score = time.perf_counter() - t0
    return {"value": round(16 / t_train, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "anomaly_maps_per_sec": round(2 / t_score, 3),
            "sample": "oracle on torch-CPU fp32: 2 timed train steps of batch 16 @256x256 (fwd+bwd+SGD); "
                      "scoring of 2 images (1682 patches, 588-row bank, blur+bilinear)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--phase", choices=["both", "train", "score"], default="both")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-precision", choices=["32", "16"], default="32",
                    help="32: exact fp32 MFMA (headline); 16: bf16-operand MFMA, the reference's Trainer(precision=16)")
    ap.add_argument("--no-bf16-extra", action="store_true", help="skip the additional precision=16 training measurement")
    ap.add_argument("--no-x3-extra", action="store_true",
                    help="skip the additional split-bf16 (bf16x3: hi*hi + hi*lo + lo*hi, fp32-class accuracy) measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
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
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from self_supervised import ops
    from self_supervised.models import AnomalyDetector

    model = build_model(dev)
    x = synth_images(args.batch, args.size, 1234 + rank, dev)
    y = torch.randint(0, 4, (args.batch,), generator=torch.Generator().manual_seed(1235 + rank)).to(dev)
    bank = torch.randn(588, 512, generator=torch.Generator().manual_seed(2)).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()  # noqa

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    res = {}
    prof = {}
    if args.phase in ("both", "train"):
        from self_supervised import training
        model.train()
        model.unfreeze()
        trainer = training.DataParallelStep(model, lr=0.005, world_size=world, precision=int(args.train_precision))
        ops.PROFILE = None
        for _ in range(args.warmup):
            trainer.step(x, y)
        ops.PROFILE = []
        dt = timed(lambda: trainer.step(x, y), args.steps, 0)
        prof["train"] = ops.drain_profile()
        res["train_s"] = dt
        # extras, never the headline: the same step with bf16-operand MFMA (what the reference's precision=16 asks for) and
        # with split-bf16 emulation of the fp32 product.  A failing extra must not take the headline line with it (on one
        # GPU; with several ranks every rank runs the same code, an exception there is fatal either way).
        extras = []
        if args.train_precision == "32" and not args.no_bf16_extra:
            extras.append(("train16_s", 16))
        if args.train_precision == "32" and not args.no_x3_extra:
            extras += [("train_x3_s", "bf16x3"), ("train_x6_s", "bf16x6")]
        for key, prec in extras:
            ops.PROFILE = None
            try:
                tx = training.DataParallelStep(model, lr=0.005, world_size=world, precision=prec)
                for _ in range(max(args.warmup, 1)):
                    tx.step(x, y)
                res[key] = timed(lambda: tx.step(x, y), args.steps, 0)
            except Exception as e:          # noqa: BLE001
                if world > 1:
                    raise
                print(f"[bench] extra {key} skipped: {e}", file=sys.stderr)
            trainer.eng.bf16 = False
    torch.cuda.empty_cache()
    if args.phase in ("both", "score"):
        model.eval(); model.enable_patch_level_mode()
        det = AnomalyDetector(patch_level=True, batch=args.batch, num_patches=841)
        det.fit_bank(bank)
        ops.PROFILE = None
        for _ in range(args.warmup):
            score_batch(model, det, x, args.size)
        ops.PROFILE = []
        dt = timed(lambda: score_batch(model, det, x, args.size), args.steps, 0)
        prof["score"] = ops.drain_profile()
        res["score_s"] = dt
        if not args.no_x3_extra:
            # extra, not the headline: the same scoring pass with split-bf16 products on the bf16 matrix cores
            ops.PROFILE = None
            for tag in ("x3", "x6"):
                os.environ["SSAD_MATH"] = "bf16" + tag
                try:
                    score_batch(model, det, x, args.size)
                    ops.PROFILE = []
                    res[f"score_{tag}_s"] = timed(lambda: score_batch(model, det, x, args.size), args.steps, 0)
                    prof[f"score_{tag}"] = ops.drain_profile()
                    ops.PROFILE = None
                except Exception as e:      # noqa: BLE001
                    if world > 1:
                        raise
                    print(f"[bench] extra score_{tag} skipped: {e}", file=sys.stderr)
            os.environ["SSAD_MATH"] = "f32"
        model.disable_patch_level_mode()
    ops.PROFILE = None

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    out = {
        "metric": "train images/sec + anomaly-maps/sec, ResNet-18 256x256 bs256",
        "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"ResNet-18 {args.size}x{args.size} bs{args.batch} self-sup train + anomaly map, "
                               f"{world}xMI355X, synthetic images (BASELINE configs[{1 if world == 1 else 2}])",
                   "images_per_gpu": args.batch, "patches_per_image": 841, "bank_rows": 588,
                   "parallelism": f"dp{world}"},
    }
    tot_s = 0.0
    if "train_s" in res:
        out["value"] = round(world * args.batch * args.steps / res["train_s"], 2)
        out["train_ms_per_step"] = round(1e3 * res["train_s"] / args.steps, 3)
        tot_s += res["train_s"]
    if "train16_s" in res:
        out["train_images_per_sec_precision16"] = round(world * args.batch * args.steps / res["train16_s"], 2)
    for tag in ("x3", "x6"):       # extras: split-bf16 emulation of the fp32 product (x6: fp32-faithful; x3: 4.6e-6 vs fp64)
        if f"train_{tag}_s" in res:
            out[f"train_images_per_sec_bf16{tag}"] = round(world * args.batch * args.steps / res[f"train_{tag}_s"], 2)
        if f"score_{tag}_s" in res:
            out[f"anomaly_maps_per_sec_bf16{tag}"] = round(world * args.batch * args.steps / res[f"score_{tag}_s"], 3)
    if "score_s" in res:
        out["anomaly_maps_per_sec"] = round(world * args.batch * args.steps / res["score_s"], 3)
        out["score_ms_per_step"] = round(1e3 * res["score_s"] / args.steps, 3)
        tot_s += res["score_s"]
        if "value" not in out:
            out["value"], out["unit"] = out["anomaly_maps_per_sec"], "anomaly-maps/sec"
            out["metric"] = "anomaly-maps/sec, ResNet-18 256x256 bs256 (scoring phase only)"
    out["ms_per_step"] = round(1e3 * tot_s / args.steps, 3)

    # roofline of the dominant kernel (conv_igemm_f32: every 3x3 / 1x1 conv and linear layer), live HIP events
    phase = "score" if "score" in prof else "train"
    recs = [r for r in prof[phase] if r["kernel"].startswith("conv_igemm_f32")] or \
           [r for r in prof[phase] if r["kernel"].startswith("conv_igemm")]
    if recs:
        t = sum(r["ms"] for r in recs) * 1e-3
        fl = sum(r["flops"] for r in recs)
        allk = sum(r["ms"] for r in prof[phase]) * 1e-3
        ach = fl / t / 1e12
        traffic = None          # HBM-side bytes per launch of this kernel from the committed PMC passes (same command line)
        tj = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if phase == "score" and args.batch == 256 and args.size == 256 and os.path.exists(tj):
            traffic = json.load(open(tj))["traffic_MB_per_launch"] * 1e6
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                           "kernel": "conv_igemm_f32_kernel", "phase": phase, "launches": len(recs),
                           "avg_launch_ms": round(1e3 * t / len(recs), 4),
                           "alg_gflop_per_launch": round(fl / len(recs) / 1e9, 3),
                           "alg_GBps": round(sum(r["bytes"] for r in recs) / t / 1e9, 1),
                           "share_of_gpu_time": round(t / allk, 4)}
        # the split-bf16 extras against the bf16 matrix-core peak: executed MFMA FLOPs = 3x / 6x the algorithmic ones
        for tag, mult in (("x3", 3), ("x6", 6)):
            rx = [r for r in prof.get(f"score_{tag}", []) if r["kernel"].startswith("conv_igemm_x")]
            if rx:
                tx = sum(r["ms"] for r in rx) * 1e-3
                alg = sum(r["flops"] for r in rx) / tx / 1e12
                out[f"roofline_bf16{tag}"] = {"bound": "mfma", "kernel": f"conv_igemm (bf16{tag})", "phase": "score",
                                              "alg_TFLOPs": round(alg, 1), "achieved": round(mult * alg, 1), "peak": 2500.0,
                                              "unit": "TFLOP/s (bf16 MFMA, executed)", "frac": round(mult * alg / 2500.0, 4)}
        out["kernel_ms"] = {}
        for ph in prof:
            if ph.startswith("score_x"):
                continue
            by = {}
            for r in prof[ph]:
                e = by.setdefault(r["kernel"], [0.0, 0, 0.0]); e[0] += r["ms"]; e[1] += 1; e[2] += r["flops"]
            out["kernel_ms"][ph] = {k: [round(v[0] / args.steps, 3), v[1] // args.steps,
                                        round(v[2] / max(v[0], 1e-9) / 1e9, 1)] for k, v in sorted(by.items())}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
