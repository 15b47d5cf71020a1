"""Does tools.training(gpu_pipeline=True) get slower call after call in one process (its sampler pool is fork()ed from the GPU
process once per stage)?  Prints the median fine-tune epoch rate of six consecutive calls."""
import contextlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("SSAD_ALLOW_RANDOM_BACKBONE", "1")


def main():
    import torch
    from fake_mvtec import make_tree
    from self_supervised import tools
    with tempfile.TemporaryDirectory() as tmp:
        root = make_tree(os.path.join(tmp, "data"), categories=("bottle",), n_train=40, n_test_good=2, n_test_bad=2, size=256)
        gpu = not (len(sys.argv) > 1 and sys.argv[1] == "cpu")       # "cpu": the reference's own DataLoader(num_workers=8) input path
        for rep in range(6 if gpu else 2):
            with contextlib.redirect_stdout(sys.stderr):
                hist = tools.training(root + "bottle/", os.path.join(tmp, f"out{rep}") + "/", "bottle", imsize=(256, 256), batch_size=96, seed=0,
                                      projection_training_params=(1, 0.03), fine_tune_params=(4, 0.005),
                                      trainer_kwargs={"limit_val_batches": 1}, gpu_pipeline=gpu)
            rates = sorted(n / t for n, t in hist["throughput"]["fine_tune"][1:])
            print("RESULT call", rep, "fine-tune img/s", [round(n / t) for n, t in hist["throughput"]["fine_tune"]], flush=True)


if __name__ == "__main__":
    main()
