"""cProfile of tools.inference on a fake MVTec category (where the host time of an evaluation goes).
   python tools/profile_inference.py  -> top cumulative entries on stdout"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("SSAD_ALLOW_RANDOM_BACKBONE", "1")
import torch
from fake_mvtec import make_tree
from self_supervised import tools


def main():
    with tempfile.TemporaryDirectory() as tmp:
        root = make_tree(os.path.join(tmp, "data"), categories=("bottle",), n_train=40, n_test_good=48, n_test_bad=48, size=256)
        out = os.path.join(tmp, "out") + "/"
        tools.training(root + "bottle/", out, "bottle", imsize=(256, 256), batch_size=32, seed=0, projection_training_params=(1, 0.03),
                       fine_tune_params=(1, 0.005), trainer_kwargs={"limit_train_batches": 2, "limit_val_batches": 1}, gpu_pipeline=True)
        ck = out + "best_model.ckpt"
        tools.inference(ck, root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)
        torch.cuda.synchronize()
        from self_supervised import trainer as _tr
        _orig = _tr._to_device
        acc = {"wait": 0.0, "copy": 0.0, "n": 0}
        def _timed(batch, dev):
            t0 = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
            out = _orig(batch, dev); torch.cuda.synchronize(); t2 = time.perf_counter()
            acc["wait"] += t1 - t0; acc["copy"] += t2 - t1; acc["n"] += 1
            return out
        _tr._to_device = _timed
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        r = tools.inference(ck, root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)
        torch.cuda.synchronize()
        pr.disable()
        print("seconds", time.perf_counter() - t0, "images", r.anomaly_maps.shape[0], "to_device: waited for the GPU", acc["wait"], "copies", acc["copy"], "calls", acc["n"])
        pstats.Stats(pr).strip_dirs().sort_stats("cumulative").print_stats(40)
        pstats.Stats(pr).strip_dirs().sort_stats("tottime").print_callers("method 'to' of")


if __name__ == "__main__":        # DataLoader workers come from a fork server once the GPU is up: the main module is re-imported there
    main()
