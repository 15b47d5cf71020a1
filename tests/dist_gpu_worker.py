"""Worker of tests/test_hip_distributed.py: one of N ranks that SHARE the box's single GPU (gloo stages the collectives
through the host -- the data-parallel code path is the same as over RCCL, only the transport differs).
Launched by `python -m torch.distributed.run`; prints one line `RESULT {...json...}` on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist


def allsame(t):
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t.contiguous())
    return all(torch.equal(parts[0], p) for p in parts), parts


def case_step(res):
    """DataParallelStep on two ranks: replicas start identical although built from different seeds; the reduced gradient
    is the SUM of the ranks' own gradients (each recomputed by a single-rank engine on the same batch); weights stay
    identical over eager + captured (hipGraph) steps; the graph plan is cut at the bucket boundaries."""
    from oracle import weights as ow
    from self_supervised import training
    from self_supervised.models import PeraNet
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    sd = ow.seeded_state_dict(rank)                      # DIFFERENT weights per rank on purpose
    m = PeraNet(); m.load_state_dict(sd); m.to(dev).train(); m.unfreeze()
    dp = training.DataParallelStep(m, lr=0.03, world_size=world)
    ok, parts = allsame(dp.eng.arena.p)
    res["broadcast_params_equal"] = ok
    bufs = torch.cat([b.float().flatten() for b in m.buffers()])
    res["broadcast_buffers_equal"] = allsame(bufs)[0]
    x = ow.synthetic_images(4, 64, seed=70 + rank).to(dev)
    y = ow.synthetic_labels(4, seed=80 + rank).to(dev)
    # this rank's own gradient from an independent single-rank engine holding the same (rank 0) weights
    m1 = PeraNet(); m1.load_state_dict({k: v.clone() for k, v in m.state_dict().items()}); m1.to(dev).train(); m1.unfreeze()
    s1 = training.DataParallelStep(m1, lr=0.03, world_size=1, graph=False)
    s1.step(x, y)
    own = s1.eng.arena.g.clone()
    _, gparts = allsame(own)
    want_sum = gparts[0] + gparts[1]
    dp.step(x, y)                                        # eager first step (captures on the second)
    got = dp.eng.arena.g
    err = (got - want_sum).abs().max().item() / max(want_sum.abs().max().item(), 1e-12)
    res["allreduce_rel_err"] = err
    res["buckets"] = len(dp.bucketer.launched)
    res["weights_equal_after_step1"] = allsame(dp.eng.arena.p)[0]
    for _ in range(3):                                   # step 2 captures + replays, 3-4 replay
        dp.step(x, y)
    torch.cuda.synchronize()
    res["weights_equal_after_graph_steps"] = allsame(dp.eng.arena.p)[0]
    res["momentum_equal"] = allsame(dp.eng.arena.m)[0]
    plan = next(iter(dp._plans.values())) if dp._plans else None
    res["graph_segments"] = sum(1 for o in plan["ops"] if o[0] == "graph") if plan else 0
    res["graph_allreduces"] = sum(1 for o in plan["ops"] if o[0] == "allreduce") if plan else 0
    # captured steps == eager steps: an eager-only twin fed the same batches ends on the same weights
    m2 = PeraNet(); m2.load_state_dict(sd); m2.to(dev).train(); m2.unfreeze()
    dp2 = training.DataParallelStep(m2, lr=0.03, world_size=world, graph=False)
    for _ in range(4):
        dp2.step(x, y)
    torch.cuda.synchronize()
    res["graph_vs_eager_max_abs"] = (dp2.eng.arena.p - dp.eng.arena.p).abs().max().item()
    res["finite"] = bool(torch.isfinite(dp.eng.arena.p).all().item())


def case_selfcheck(res):
    """DataParallelStep.self_check on two ranks: a healthy job passes, keeps its verified plan and its state untouched; a replay
    that goes wrong on ONE rank (a parameter perturbed after the replayed step) sends BOTH ranks to eager launches."""
    from oracle import weights as ow
    from self_supervised import training
    from self_supervised.models import PeraNet
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    x = ow.synthetic_images(4, 64, seed=70 + rank).to(dev)
    y = ow.synthetic_labels(4, seed=80 + rank).to(dev)
    m = PeraNet(); m.load_state_dict(ow.seeded_state_dict(rank)); m.to(dev).train(); m.unfreeze()
    dp = training.DataParallelStep(m, lr=0.03, world_size=world)
    before = torch.cat([dp.eng.arena.p, dp.eng.arena.m] + [b.detach().flatten().float() for b in m.buffers()]).clone()
    rep = dp.self_check(x, y)
    after = torch.cat([dp.eng.arena.p, dp.eng.arena.m] + [b.detach().flatten().float() for b in m.buffers()])
    res["healthy"] = rep
    res["state_restored"] = bool(torch.equal(before, after))
    res["plan_kept"] = len(dp._plans) == 1
    for _ in range(3):
        dp.step(x, y)
    torch.cuda.synchronize()
    res["weights_equal_after_steps"] = allsame(dp.eng.arena.p)[0]
    # fault injection: the replay leaves a different bit pattern on rank 1 only
    m2 = PeraNet(); m2.load_state_dict(ow.seeded_state_dict(7)); m2.to(dev).train(); m2.unfreeze()
    bad = training.DataParallelStep(m2, lr=0.03, world_size=world)
    orig = bad._replay

    def broken_replay(plan, xx, yy):
        out = orig(plan, xx, yy)
        if rank == 1:
            torch.cuda.synchronize()
            bad.eng.arena.p.view(torch.int32)[17] ^= 1
        return out
    bad._replay = broken_replay
    rep2 = bad.self_check(x, y)
    res["faulty"] = rep2
    res["faulty_use_graph"] = bool(bad.use_graph)
    bad._replay = orig
    for _ in range(3):
        bad.step(x, y)
    torch.cuda.synchronize()
    res["faulty_plans"] = len(bad._plans)
    res["faulty_weights_equal"] = allsame(bad.eng.arena.p)[0]
    # a capture the runtime refuses on ONE rank only: no rank may go on to replay (its per-bucket all-reduces would meet the
    # other rank's different collective); both fall back to eager launches together and keep training
    m3 = PeraNet(); m3.load_state_dict(ow.seeded_state_dict(9)); m3.to(dev).train(); m3.unfreeze()
    half = training.DataParallelStep(m3, lr=0.03, world_size=world)
    orig_cap = half._capture

    def one_sided_capture(xx, yy, key):
        if rank == 1:
            raise RuntimeError("injected: capture refused on rank 1")
        return orig_cap(xx, yy, key)
    half._capture = one_sided_capture
    rep3 = half.self_check(x, y)
    half._capture = orig_cap
    res["one_sided"] = rep3
    res["one_sided_use_graph"] = bool(half.use_graph)
    for _ in range(3):
        half.step(x, y)
    torch.cuda.synchronize()
    res["one_sided_weights_equal"] = allsame(half.eng.arena.p)[0]


def case_rccl1(res):
    """The transport the gloo cases cannot reach: a ONE-rank RCCL communicator on the box's card, with the step told it
    has two replicas (world_size=2), so every bucket goes through ProcessGroupNCCL's stream / event hand-over between the
    hipGraph segments.  A one-rank sum is the identity, so the weights must equal an eager single-rank run whose SGD
    kernel applies the same 1/2 factor -- bit for bit, eager and replayed."""
    from oracle import weights as ow
    from self_supervised import training
    from self_supervised.models import PeraNet
    dev = torch.device("cuda", 0)
    sd = ow.seeded_state_dict(3)
    x = ow.synthetic_images(4, 64, seed=71).to(dev)
    y = ow.synthetic_labels(4, seed=81).to(dev)
    m = PeraNet(); m.load_state_dict(sd); m.to(dev).train(); m.unfreeze()
    dp = training.DataParallelStep(m, lr=0.03, world_size=2)
    m1 = PeraNet(); m1.load_state_dict(sd); m1.to(dev).train(); m1.unfreeze()
    s1 = training.DataParallelStep(m1, lr=0.03, world_size=1, graph=False)
    s1.opt.grad_scale = 0.5
    losses = []
    for i in range(5):                                   # step 1 eager, step 2 captures, 3-5 replay
        a = dp.step(x, y); b = s1.step(x, y)
        losses.append((float(a[0]), float(b[0])))
        if i == 0:
            res["buckets"] = len(dp.bucketer.launched)
            res["eager_max_abs"] = (dp.eng.arena.p - s1.eng.arena.p).abs().max().item()
    torch.cuda.synchronize()
    plan = next(iter(dp._plans.values())) if dp._plans else None
    res["graph_segments"] = sum(1 for o in plan["ops"] if o[0] == "graph") if plan else 0
    res["graph_allreduces"] = sum(1 for o in plan["ops"] if o[0] == "allreduce") if plan else 0
    res["replay_max_abs"] = (dp.eng.arena.p - s1.eng.arena.p).abs().max().item()
    res["momentum_max_abs"] = (dp.eng.arena.m - s1.eng.arena.m).abs().max().item()
    res["losses_equal"] = all(a == b for a, b in losses)
    res["finite"] = bool(torch.isfinite(dp.eng.arena.p).all().item())
    res["backend"] = dist.get_backend()


def case_fit(res, tmp):
    """tools.training on two ranks: identical parameters on both ranks at the end, one loadable checkpoint."""
    from fake_mvtec import make_tree
    from self_supervised import tools, datasets
    from self_supervised.models import PeraNet
    datasets._DataModule.num_workers = 0
    rank = dist.get_rank()
    root = os.path.join(tmp, "data")
    if rank == 0:
        make_tree(root, categories=("bottle",), n_train=16, n_test_good=2, n_test_bad=2, size=80)
    dist.barrier()
    out = os.path.join(tmp, "out") + "/"
    captured = {}
    orig_fit = tools.Trainer.fit

    def fit(self, model, *a, **k):
        orig_fit(self, model, *a, **k)
        captured["model"] = model
    tools.Trainer.fit = fit
    hist = tools.training(root + "/bottle/", out, "bottle", imsize=(64, 64), batch_size=4, seed=0,
                          projection_training_params=(1, 0.03), fine_tune_params=(2, 0.005),
                          trainer_kwargs={"limit_train_batches": 3, "limit_val_batches": 1})
    m = captured["model"]
    pflat = torch.cat([p.detach().flatten().float() for p in m.parameters()])
    res["fit_params_equal"] = allsame(pflat.cuda())[0]
    res["fit_bank_equal"] = allsame(torch.tensor([float(m.memory_bank.shape[0])]).cuda())[0]
    res["ckpt_exists"] = os.path.exists(out + "best_model.ckpt")
    m2 = PeraNet.load_from_checkpoint(out + "best_model.ckpt")
    res["ckpt_loads"] = m2.stage == "fine_tune"
    res["hist_ok"] = len(hist["fine_tune"]["train"]["loss"]) == 2


def main():
    case, tmp = sys.argv[1], sys.argv[2]
    os.environ.setdefault("SSAD_ALLOW_RANDOM_BACKBONE", "1")
    torch.cuda.set_device(0)
    if case == "rccl1":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    res = {}
    if case == "step":
        case_step(res)
    elif case == "selfcheck":
        case_selfcheck(res)
    elif case == "rccl1":
        case_rccl1(res)
    else:
        case_fit(res, tmp)
    if dist.get_rank() == 0:
        print("RESULT " + json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
