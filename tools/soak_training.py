"""tools.training with ALL its defaults (10 + 30 epochs, batch 96, precision 16, GPU pipeline) and tools.inference on two synthetic
categories -- one with a fixed object mask (bottle), one whose masks are built per image on the device (screw): wall time, median
fine-tune epoch, the final validation loss, left-over shared-memory files, peak device memory.
   python tools/soak_training.py        (round 4 on one MI355X: 6-10 s per category, 0.097 s per epoch of 960 images)"""
import os, sys, tempfile, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("SSAD_ALLOW_RANDOM_BACKBONE", "1")
import torch
from fake_mvtec import make_tree
from self_supervised import tools

def main():
    with tempfile.TemporaryDirectory() as tmp:
        root = make_tree(os.path.join(tmp, "data"), categories=("bottle", "screw"), n_train=40, n_test_good=8, n_test_bad=8, size=256)
        for sub in ("bottle", "screw"):
            t0 = time.perf_counter()
            hist = tools.training(root + sub + "/", os.path.join(tmp, "out_" + sub) + "/", sub, imsize=(256, 256), batch_size=96, seed=0)   # all defaults: 30 + 20 epochs, precision 16
            dt = time.perf_counter() - t0
            ft = hist["throughput"]["fine_tune"]
            print(sub, "epochs", len(hist["throughput"]["projection_train"]), len(ft), "wall", round(dt, 1), "s; fine-tune median epoch",
                  round(sorted(t for _, t in ft)[len(ft) // 2], 4), "final val_loss", hist["fine_tune"]["val"]["loss"][-1], flush=True)
            r = tools.inference(os.path.join(tmp, "out_" + sub) + "/best_model.ckpt", root + sub + "/", sub, mvtec_inference=True, patch_localization=True)
            print(sub, "maps", tuple(r.anomaly_maps.shape), "finite", bool(torch.isfinite(r.anomaly_maps).all()), flush=True)
    print("shm leftovers:", [f for f in glob.glob("/dev/shm/*") if "ssad" in f])
    print("max GPU memory GB", round(torch.cuda.max_memory_allocated() / 1e9, 2))

if __name__ == "__main__":
    main()
