"""cProfile of tools.training with its default arguments on a fake MVTec category (where the host time of an epoch goes).
   python tools/profile_training.py [precision]  -> epoch times + top cumulative entries on stdout"""
import cProfile
import os
import pstats
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("SSAD_ALLOW_RANDOM_BACKBONE", "1")
import torch
from fake_mvtec import make_tree
from self_supervised import tools


def main():
    prec = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    with tempfile.TemporaryDirectory() as tmp:
        root = make_tree(os.path.join(tmp, "data"), categories=("bottle",), n_train=40, n_test_good=8, n_test_bad=8, size=256)
        kw = dict(imsize=(256, 256), batch_size=96, seed=0, projection_training_params=(1, 0.03),
                  trainer_kwargs={"precision": prec, "limit_val_batches": 1})
        tools.training(root + "bottle/", os.path.join(tmp, "warm") + "/", "bottle", fine_tune_params=(2, 0.005), **kw)     # warm-up: pool, plans
        pr = cProfile.Profile()
        pr.enable()
        hist = tools.training(root + "bottle/", os.path.join(tmp, "out") + "/", "bottle", fine_tune_params=(6, 0.005), **kw)
        torch.cuda.synchronize()
        pr.disable()
        print("fine-tune epochs (images, seconds):", [(n, round(t, 4)) for n, t in hist["throughput"]["fine_tune"]])
        pstats.Stats(pr).strip_dirs().sort_stats("cumulative").print_stats(70)


if __name__ == "__main__":
    main()
