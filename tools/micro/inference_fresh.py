"""tools.inference in a FRESH process (argv[1] = 'train': make a fake category + checkpoint under /tmp/ssad_inf; 'infer': time it)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("SSAD_ALLOW_RANDOM_BACKBONE", "1")
import torch
from fake_mvtec import make_tree
from self_supervised import tools
tmp = "/tmp/ssad_inf"


def main():
    if sys.argv[1] == "train":
        root = make_tree(os.path.join(tmp, "data"), categories=("bottle",), n_train=40, n_test_good=12, n_test_bad=12, size=256)
        tools.training(root + "bottle/", tmp + "/out/", "bottle", imsize=(256, 256), batch_size=32, seed=0, projection_training_params=(1, 0.03),
                       fine_tune_params=(1, 0.005), trainer_kwargs={"limit_train_batches": 2, "limit_val_batches": 1}, gpu_pipeline=True)
    else:
        root = tmp + "/data/"
        import cProfile, pstats, gc
        if len(sys.argv) > 2:
            from self_supervised import datasets
            datasets._DataModule.num_workers = int(sys.argv[2])
        for rep in range(3):
            pr = cProfile.Profile()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pr.enable()
            r = tools.inference(tmp + "/out/best_model.ckpt", root + "bottle/", "bottle", mvtec_inference=True, patch_localization=True)
            torch.cuda.synchronize()
            pr.disable()
            print("RESULT inference", rep, round(time.perf_counter() - t0, 3), "s for", r.anomaly_maps.shape[0], "images",
                  "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 2), "allocated GB", round(torch.cuda.memory_allocated() / 2**30, 2))
            pstats.Stats(pr).strip_dirs().sort_stats("tottime").print_stats(8)


if __name__ == "__main__":
    main()
