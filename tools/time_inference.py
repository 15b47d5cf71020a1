"""Wall time of tools.inference + tools.upsample on the bench's synthetic category (96 test PNGs of 256 x 256), five calls after a warm-up,
with the time stamps of its phases (SSAD_TIMELINE=1 makes tools.inference record them).  python tools/time_inference.py [size]"""
import gc
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("SSAD_ALLOW_RANDOM_BACKBONE", "1")
os.environ["SSAD_TIMELINE"] = "1"
import contextlib
import torch
from fake_mvtec import make_tree
from self_supervised import tools
from self_supervised.models import PeraNet


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    with tempfile.TemporaryDirectory() as tmp:
        root = make_tree(os.path.join(tmp, "data"), categories=("bottle",), n_train=40, n_test_good=48, n_test_bad=48, size=size)
        torch.manual_seed(0)
        m = PeraNet()
        ck = os.path.join(tmp, "m.ckpt")
        torch.save({"state_dict": m.state_dict(), "hyper_parameters": {}, "memory_bank": torch.tensor([]),
                    "optimizer_states": [{"momentum": torch.zeros(12691524)}]}, ck)
        with contextlib.redirect_stdout(sys.stderr):
            tools.inference(ck, root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)
        torch.cuda.synchronize()
        gc.collect(); gc.disable()
        for i in range(5):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(sys.stderr):
                r = tools.inference(ck, root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)
                up = tools.upsample(r.anomaly_maps, 256, verbose=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tl = getattr(tools, "TIMELINE", [])
            print(f"call {i}: {dt * 1e3:.1f} ms = {up.shape[0] / dt:.1f} maps/s | " +
                  "  ".join(f"{k} {1e3 * (t - t0):.0f}" for k, t in tl))


if __name__ == "__main__":
    main()
