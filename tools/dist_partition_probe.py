#!/usr/bin/env python3
"""Two ranks on one GPU (gloo): per-step time when the batch shape changes (256 -> 128 -> 256) under data parallel replay.
   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 tools/dist_partition_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import torch, torch.distributed as dist
os.environ.setdefault("SSAD_ALLOW_RANDOM_BACKBONE", "1")
from self_supervised import training
from self_supervised.models import PeraNet
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("gloo")
torch.manual_seed(rank)
m = PeraNet().to(dev).train(); m.unfreeze()
st = training.DataParallelStep(m, lr=0.005, world_size=world, precision=32)
if os.environ.get("PROBE_TRACE"):
    def traced(plan, x, y):
        if x.data_ptr() != plan["x"].data_ptr():
            plan["x"].copy_(x, non_blocking=True)
        if y.data_ptr() != plan["y"].data_ptr():
            plan["y"].copy_(y, non_blocking=True)
        works, g, log = [], st.eng.arena.g, []
        for op in plan["ops"]:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if op[0] == "graph":
                op[1].replay(); torch.cuda.synchronize()
            elif op[0] == "allreduce":
                works.append(dist.all_reduce(g[op[1]:op[2]], op=dist.ReduceOp.SUM, group=st.pg, async_op=True))
            else:
                for w in works:
                    w.wait()
                works = []
            log.append(f"{op[0][:2]}{(time.perf_counter() - t0) * 1e3:.0f}")
        if rank == 0:
            print("   ", " ".join(log), flush=True)
        return plan["out"]
    st._replay = traced
for B in (256, 128, 256, 64):
    x = torch.randn(B, 3, 256, 256, device=dev); y = torch.randint(0, 4, (B,), device=dev)
    ts = []
    for i in range(6):
        torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
        st.step(x, y); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    if rank == 0:
        print(B, " ".join(f"{t:8.1f}" for t in ts), flush=True)
dist.destroy_process_group()
