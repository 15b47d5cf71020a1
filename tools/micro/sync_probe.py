"""Why does torch.cuda.synchronize() take ~140 ms inside tools.inference's predict loop?  Probe: iterate the MVTec predict loader
(8 forked workers) with NO GPU work queued and time synchronize / a tiny H2D copy per batch."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from fake_mvtec import make_tree
from self_supervised.datasets import MVTecDatamodule
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
with tempfile.TemporaryDirectory() as tmp:
    root = make_tree(os.path.join(tmp, "data"), categories=("bottle",), n_train=8, n_test_good=12, n_test_bad=12, size=256)
    for nw in (8, 0):
        dm = MVTecDatamodule(root + "bottle/", batch_size=1)
        dm.num_workers = nw
        dm.setup("predict")
        dl = dm.predict_dataloader()
        print("workers", getattr(dl, "num_workers", None))
        ts, tc, tl = [], [], []
        t_prev = time.perf_counter()
        for b in dl:
            t0 = time.perf_counter(); tl.append(t0 - t_prev)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            x = [u.to(dev, non_blocking=True) for u in b]; torch.cuda.synchronize(); t2 = time.perf_counter()
            ts.append(t1 - t0); tc.append(t2 - t1); t_prev = time.perf_counter()
        f = lambda v: [round(1e3 * u, 2) for u in v[:12]]
        print(" loader wait ms", f(tl)); print(" synchronize ms", f(ts)); print(" copy ms", f(tc))
