#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference; the GPU box never has it).
The reference's own glue (PeraNet.forward / training_step / predict_step label logic,
AnomalyDetector, extract_patches, gt2label, dataset_generator primitives) is imported
as-is; third-party packages that are missing here are replaced by throw-away
``sys.modules`` stubs:

  torchsummary                      -> no-op
  pytorch_lightning.LightningModule -> nn.Module + no-op save_hyperparameters/log_dict
  torchmetrics.functional.accuracy  -> argmax accuracy
  torchvision.models.resnet18       -> oracle.resnet18 (restated from the public spec)
  skimage / torchvision.transforms  -> empty shells (only names are imported)

Outputs are data only (inputs are regenerated from seeds by oracle.weights):
  forward.npz, train_step.npz, detector.npz, patches.npz, upsample.npz, cutpaste.npz, auroc.npz, gradcam.npz,
  getitem.npz, metrics.npz

    python tests/golden/make_fixtures.py            # everything
    python tests/golden/make_fixtures.py gradcam    # one section
    python tests/golden/make_fixtures.py getitem    # PretextTaskDataset.__getitem__ of the reference (datasets.py:209-394)
    python tests/golden/make_fixtures.py metrics    # the reference's metrics.py + tools.Evaluator on seeded maps

The `getitem` section imports the reference's datasets.py / dataset_generator.py and runs ITS __getitem__ on the synthetic
MVTec-shaped tree of tests/fake_mvtec.py under fixed python / numpy / torch seeds.  Third-party pieces that are absent here
are stand-ins, flagged: torchvision.transforms := self_supervised/tv_transforms.py of this repo (restated from
torchvision's public behaviour, RNG calls in torchvision's order); skimage.feature.canny := this repo's Canny restatement (identical to scikit-image 0.18.3's: tests/golden/skimage.npz);
skimage.morphology.square / label := numpy / scipy.ndimage equivalents (label with skimage's default full connectivity).
Everything else -- label draw, affine / crop order, defect source choice, generate_patch, colour-similarity brightness
jitter, container clamp, rect2poly, scar rotation and pasting, poly-line sampling + savgol + ImageDraw.line, jitter call
order, obj_mask's morphology chain -- is the reference's own code running on PIL / numpy / scipy.
"""
import os
import random
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SRC = "/root/reference/src"

from oracle import weights as ow            # noqa: E402
from oracle import scoring as osc           # noqa: E402
from oracle import resnet18 as ores         # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    _stub("torchsummary", summary=lambda *a, **k: None)

    class LightningModule(nn.Module):
        current_epoch = 0

        def save_hyperparameters(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    class _Any:
        def __init__(self, *a, **k):
            pass

    pl = _stub("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=type("LightningDataModule", (), {}),
               Trainer=_Any, Callback=_Any)
    _stub("pytorch_lightning.callbacks", ModelCheckpoint=_Any, Callback=_Any)
    pl.callbacks = sys.modules["pytorch_lightning.callbacks"]

    def accuracy(y_hat, y):
        return (y_hat.argmax(1) == y).float().mean()

    tm = _stub("torchmetrics", JaccardIndex=_Any, PrecisionRecallCurve=_Any, F1Score=_Any)
    tm.functional = _stub("torchmetrics.functional", accuracy=accuracy)

    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models", resnet18=ores.resnet18, ResNet=ores.ResNet18)
    tv.transforms = _stub("torchvision.transforms", ColorJitter=_Any, Compose=_Any)
    tv.transforms.functional = _stub("torchvision.transforms.functional")

    sk = _stub("skimage")
    sk.morphology = _stub("skimage.morphology", square=None, label=None)
    sk.feature = _stub("skimage.feature")
    sk.segmentation = _stub("skimage.segmentation", slic=None)
    sk.color = _stub("skimage.color")
    sk.transform = _stub("skimage.transform", swirl=None)


def t2n(t):
    return t.detach().cpu().numpy().copy()


def make_gradcam(rm, sd):
    """(viii) Grad-CAM (gradcam.py:25-48): the reference class on the seeded model, one image per call."""
    import warnings
    from self_supervised import gradcam as rgc
    model = rm.PeraNet()
    model.load_state_dict(sd, strict=True)
    cam = rgc.GradCam(model)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")               # register_backward_hook deprecation
        x = ow.synthetic_images(2, 64, seed=301)
        out["cam64_auto"] = t2n(cam(x[0:1]))          # class_idx None -> arg-max logit
        out["cam64_c1"] = t2n(cam(x[1:2], 1))
        out["cam64_c2"] = t2n(cam(x[1:2], torch.tensor(2)))
        x128 = ow.synthetic_images(1, 128, seed=302)
        out["cam128_auto"] = t2n(cam(x128))
        x32 = ow.synthetic_images(1, 32, seed=303)    # below 64 px: nearest-resize branch, saliency back at 32x32
        out["cam32_c1"] = t2n(cam(x32, 1))
    np.savez_compressed(os.path.join(HERE, "gradcam.npz"), **out)


GETITEM_CASES = [("bottle", False), ("bottle", True), ("carpet", False), ("carpet", True), ("capsule", True), ("screw", True),
                 ("screw", False)]
GETITEM_SIZE = {("capsule", True): (256, 64), ("screw", True): (256, 64)}      # (imsize, patch_size); default (64, 32): the fixed
                                                                                # pre-crops of datasets.py:244-249 presume 256 px
GETITEM_TREE = ("bottle", "carpet", "capsule", "screw")
GETITEM_SAMPLES = 12            # seeds 0..11 per case; every label 0..3 and every defect-source branch occurs


def make_getitem():
    """(ix) PretextTaskDataset.__getitem__ of the reference on a synthetic tree (see the module docstring)."""
    import importlib.util
    import tempfile
    from scipy import ndimage
    pkg = os.path.join(ROOT, "self-supervised-anomaly-detection_amd", "self_supervised")

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    tvt = load("torchvision.transforms", os.path.join(pkg, "tv_transforms.py"))
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision"].transforms = tvt
    own_gen = load("_own_dataset_generator", os.path.join(pkg, "dataset_generator.py"))       # only for its Canny restatement
    sk = sys.modules["skimage"]
    sk.feature.canny = lambda gray, sigma, low_threshold, high_threshold: own_gen._canny(gray, sigma, low_threshold, high_threshold)
    sk.morphology.square = lambda n: np.ones((n, n), int)
    sk.morphology.label = lambda a: ndimage.label(a, structure=np.ones((3, 3), int))[0]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_mvtec import make_tree
    from self_supervised import datasets as rd, constants as rconst          # the REFERENCE's modules
    assert rd.__file__.startswith(REF_SRC)
    rgen = sys.modules["self_supervised.dataset_generator"]                 # may have been imported before the stubs were filled in
    for name, fn in (("canny", sk.feature.canny), ("square", sk.morphology.square), ("label", sk.morphology.label)):
        if hasattr(rgen, name):
            setattr(rgen, name, fn)
    out = {"cases": np.array([f"{s}:{int(p)}" for s, p in GETITEM_CASES]), "n_samples": np.int64(GETITEM_SAMPLES)}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        make_tree(os.path.join(tmp, "dataset"), categories=GETITEM_TREE, n_train=4, n_test_good=1, n_test_bad=1, size=96)
        os.chdir(tmp)                                    # the reference hard-codes 'dataset/' (datasets.py:189-200)
        try:
            for subject, patch in GETITEM_CASES:
                names = np.array(sorted(os.path.join("dataset", subject, "train/good", f) for f in
                                        os.listdir(os.path.join("dataset", subject, "train/good"))))
                size, ps = GETITEM_SIZE.get((subject, patch), (64, 32))
                ds = rd.PretextTaskDataset(subject, names, imsize=(size, size), transform=None, patch_localization=patch, patch_size=ps)
                out[f"{subject}_{int(patch)}_size"] = np.array([size, ps])
                key = f"{subject}_{int(patch)}"
                out[key + "_seg"] = np.array(ds.fixed_segmentation.convert("1"))
                out["tree"] = np.array(GETITEM_TREE)
                xs, ys = [], []
                for s in range(GETITEM_SAMPLES):
                    random.seed(s); np.random.seed(s); torch.manual_seed(s)
                    x, y, orig = ds[s % len(names)]
                    xs.append(np.array(x)); ys.append(y)
                    if s == 0:
                        out[key + "_orig0"] = (orig * 255).round().byte().numpy()
                out[key + "_x"], out[key + "_y"] = np.stack(xs), np.array(ys)
                # the RNG streams after the last sample: any extra / missing draw anywhere shows up here
                out[key + "_rng"] = np.array([random.random(), np.random.rand(), float(torch.rand(1))])
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "getitem.npz"), **out)
    print("getitem.npz: labels", {k: out[k].tolist() for k in out if k.endswith("_y")})


def install_torchmetrics_spec_stub():
    """torchmetrics is absent here.  The three classes the reference's Evaluator uses (tools.py:131-146, metrics.py:42-46) are
    restated from the published torchmetrics 0.8-0.10 behaviour (the API the reference calls: PrecisionRecallCurve() without
    `task`, JaccardIndex(2, threshold=), F1Score(threshold=)), in torch and in its dtypes: third-party arithmetic pinned to the
    spec only; the reference's OWN code around them (F1 formula, argmax, indexing, flattening, target binarisation) runs for real."""
    F = torch.nn.functional

    class PrecisionRecallCurve:
        def __call__(self, preds, target):
            preds, target = preds.flatten(), target.flatten()
            order = torch.argsort(preds, descending=True)
            preds, target = preds[order], target[order]
            distinct = torch.where(preds[1:] - preds[:-1])[0]
            idx = F.pad(distinct, [0, 1], value=target.size(0) - 1)
            target = (target == 1).to(torch.long)
            tps = torch.cumsum(target * 1.0, dim=0)[idx]
            fps = 1 + idx - tps
            thr = preds[idx]
            precision = tps / (tps + fps)
            recall = tps / tps[-1]
            last = torch.where(tps == tps[-1])[0][0]
            sl = slice(0, last.item() + 1)
            precision = torch.cat([reversed(precision[sl]), torch.ones(1, dtype=precision.dtype)])
            recall = torch.cat([reversed(recall[sl]), torch.zeros(1, dtype=recall.dtype)])
            return precision, recall, reversed(thr[sl]).detach().clone()

    class JaccardIndex:
        def __init__(self, num_classes, threshold=0.5):
            self.n, self.t = num_classes, threshold

        def __call__(self, preds, target):
            p = (preds.flatten() >= self.t).long()
            t = target.flatten().long()
            conf = torch.bincount(t * self.n + p, minlength=self.n * self.n).reshape(self.n, self.n)
            inter = torch.diag(conf)
            union = conf.sum(0) + conf.sum(1) - inter
            scores = inter.float() / union.float()
            scores[union == 0] = 0.0                       # absent_score
            return scores.mean()

    class F1Score:
        def __init__(self, threshold=0.5):
            self.t = threshold

        def __call__(self, preds, target):
            if target.is_floating_point():
                raise ValueError("The `target` has to be an integer tensor.")
            p = (preds.flatten() >= self.t).int()
            t = target.flatten().int()
            tp = ((p == 1) & (t == 1)).sum()
            fp = ((p == 1) & (t == 0)).sum()
            fn = ((p == 0) & (t == 1)).sum()

            def safe(a, b):
                b = b.float().clone()
                b[b == 0] = 1.0
                return a.float() / b
            prec, rec = safe(tp, tp + fp), safe(tp, tp + fn)
            den = prec + rec
            den = torch.where(den == 0, torch.ones_like(den), den)
            return 2 * prec * rec / den

    tm = sys.modules["torchmetrics"]
    tm.PrecisionRecallCurve, tm.JaccardIndex, tm.F1Score = PrecisionRecallCurve, JaccardIndex, F1Score


def metric_cases():
    """Seeded (maps, ground truths) for the metric fixtures: blobs of several sizes, diagonal contacts (8-connectivity), a
    defect-free image, quantised scores (ties: the keep-last-of-a-run rule) and continuous ones."""
    g = torch.Generator().manual_seed(2024)
    cases = {}
    for name, n, hw, quant in (("ties", 5, 24, 32), ("smooth", 6, 32, 0)):
        gt = torch.zeros(n, 1, hw, hw)
        for i in range(n - 1):                          # the last image stays defect-free
            for _ in range(1 + i % 3):
                y, x = (int(v) for v in torch.randint(0, hw - 6, (2,), generator=g))
                h, w = (int(v) for v in torch.randint(2, 7, (2,), generator=g))
                gt[i, 0, y:y + h, x:x + w] = 1.0
            gt[i, 0, 0, 0] = 1.0; gt[i, 0, 1, 1] = 1.0   # two pixels touching by a corner: ONE region under the 3x3 structure
        maps = torch.rand(n, 1, hw, hw, generator=g) * 0.6 + 0.35 * gt * torch.rand(n, 1, hw, hw, generator=g)
        if quant:
            maps = torch.round(maps * quant) / quant
        cases[name] = (maps.float(), gt)
    return cases


def make_metrics():
    """(x) evaluation metrics: the reference's own metrics.py (compute_roc / compute_auc / compute_pro / compute_aupro /
    trapezoid / compute_f1) and tools.Evaluator (evaluate, _get_threshold) on seeded maps (metrics.py:42-228, tools.py:52-146)."""
    install_torchmetrics_spec_stub()
    _stub("self_supervised.visualization", plot_curve=lambda *a, **k: None, plot_tsne=lambda *a, **k: None)
    _stub("cv2"); _stub("seaborn")
    import contextlib
    import io
    from self_supervised import metrics as rmtr                   # the REFERENCE's modules
    from self_supervised import tools as rtools
    from self_supervised.constants import ModelOutputsContainer
    assert rmtr.__file__.startswith(REF_SRC) and rtools.__file__.startswith(REF_SRC)
    out = {"cases": np.array(sorted(metric_cases()))}
    for name, (maps, gt) in metric_cases().items():
        with contextlib.redirect_stdout(io.StringIO()):
            out[name + "_maps"], out[name + "_gts"] = t2n(maps), t2n(gt).astype(np.uint8)
            fprs, pros = rmtr.compute_pro(maps.squeeze().numpy(), gt.squeeze().numpy())
            out[name + "_fprs"], out[name + "_pros"] = fprs, pros
            out[name + "_aupro03"] = np.float64(rmtr.compute_aupro(fprs, pros, 0.3))
            out[name + "_aupro_full"] = np.float64(rmtr.compute_aupro(fprs, pros, 1.0))
            x_in = float(fprs[len(fprs) // 3])                     # an x_max that IS a curve point / one that is not
            out[name + "_trap_xs"] = np.array([x_in, 0.3, 0.123456])
            out[name + "_trap"] = np.array([rmtr.trapezoid(fprs, pros, x_max=v) for v in out[name + "_trap_xs"]] +
                                           [rmtr.trapezoid(fprs, pros)])
            flat_s, flat_t = torch.flatten(maps, 0, -1), torch.flatten(gt, 0, -1)
            fpr, tpr, thr = rmtr.compute_roc(flat_t, flat_s)
            out[name + "_roc_fpr"], out[name + "_roc_tpr"] = fpr, tpr
            out[name + "_auc"] = np.float64(rmtr.compute_auc(fpr, tpr))
            ev = rtools.Evaluator(evaluation_metrics=['auroc', 'aupro', 'iou'])
            thr_v = ev._get_threshold(flat_s, flat_t)
            out[name + "_threshold"] = np.float64(float(thr_v))
            out[name + "_f1"] = np.float64(rmtr.compute_f1(flat_t.long(), flat_s, thr_v))
            box = ModelOutputsContainer()
            box.anomaly_maps, box.ground_truths = maps.clone(), gt.clone()
            box.y_true_binary_labels = (gt.flatten(1).sum(1) > 0).long()
            ev.evaluate(box, name, None, patch_level=True)         # the reference's own call (test_patch_level_evaluation.py)
            out[name + "_eval_pixel"] = np.array([ev.scores.auroc, ev.scores.aupro, ev.scores.iou], dtype=np.float64)
    # image level (tools.py:87-98): one score and one label per image, auroc only (tools.py:101-104 exits on 'f1-score' when
    # patch_level is False); threshold and F1 through the reference's _get_threshold / compute_f1 called directly
    g = torch.Generator().manual_seed(77)
    labels = (torch.rand(60, generator=g) > 0.6).long()
    scores = (torch.rand(60, generator=g) * 0.5 + 0.25 * labels * torch.rand(60, generator=g)).float()
    scores[::7] = scores[3]                                        # ties
    with contextlib.redirect_stdout(io.StringIO()):
        box2 = ModelOutputsContainer()
        box2.anomaly_maps, box2.y_true_binary_labels = scores.clone(), labels.clone()
        ev2 = rtools.Evaluator(evaluation_metrics=['auroc'])
        ev2.evaluate(box2, "image", None)
        out["image_scores"], out["image_labels"] = t2n(scores), t2n(labels)
        out["image_auroc"] = np.float64(ev2.scores.auroc)
        thr_v = ev2._get_threshold(scores, labels)
        out["image_threshold"] = np.float64(float(thr_v))
        out["image_f1"] = np.float64(rmtr.compute_f1(labels, scores, thr_v))
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    print("metrics.npz:", {k: (v.tolist() if v.size < 5 else v.shape) for k, v in out.items() if not k.endswith(("fprs", "pros", "fpr", "tpr"))})


def main():
    assert os.path.isdir(REF_SRC), "reference not present: fixtures can only be made in the build container"
    install_stubs()
    sys.path.insert(0, REF_SRC)
    if sys.argv[1:] == ["getitem"]:
        make_getitem()
        return
    if sys.argv[1:] == ["metrics"]:
        make_metrics()
        return
    from self_supervised import models as rm                     # reference
    if sys.argv[1:] == ["gradcam"]:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        make_gradcam(rm, ow.seeded_state_dict(0))
        return
    from self_supervised import functional as rf
    from self_supervised import converters as rc
    from self_supervised import dataset_generator as rg

    torch.manual_seed(0)
    torch.set_num_threads(8)
    sd = ow.seeded_state_dict(0)

    # ---------------- (i) extract_patches ----------------
    a = torch.arange(2 * 3 * 48 * 40, dtype=torch.float32).reshape(2, 3, 48, 40)
    p_small = rf.extract_patches(a, dim=32, stride=8)
    big = torch.arange(3 * 256 * 256, dtype=torch.float32).reshape(1, 3, 256, 256)
    p_big = rf.extract_patches(big, dim=32, stride=8)
    sel = [0, 1, 28, 29, 420, 840]
    np.savez_compressed(os.path.join(HERE, "patches.npz"),
                        small=t2n(p_small).astype(np.int32),
                        big_shape=np.array(p_big.shape), big_sel=np.array(sel),
                        big_corners=np.stack([t2n(p_big[0, s, :, [0, 0, 31, 31], [0, 31, 0, 31]]) for s in sel]).astype(np.int32))

    # ---------------- (ii) forward, eval ----------------
    model = rm.PeraNet()
    missing = model.load_state_dict(sd, strict=True)
    model.eval()
    out = {}
    with torch.no_grad():
        x_img = ow.synthetic_images(2, 256, seed=1234)
        o = model(x_img)
        out["img_logits"], out["img_emb"] = t2n(o["classifier"]), t2n(o["latent_space"])
        x_c1 = ow.synthetic_images(8, 64, seed=77)
        o = model(x_c1)
        out["c1_logits"], out["c1_emb"] = t2n(o["classifier"]), t2n(o["latent_space"])
        x_32 = ow.synthetic_images(4, 32, seed=78)          # nearest-upsample branch (h < 64)
        o = model(x_32)
        out["up_logits"], out["up_emb"] = t2n(o["classifier"]), t2n(o["latent_space"])
        x_48 = ow.synthetic_images(2, 48, seed=79)          # non-integer nearest ratio 48 -> 64
        o = model(x_48)
        out["up48_logits"], out["up48_emb"] = t2n(o["classifier"]), t2n(o["latent_space"])
        # patch level: one 256^2 image -> 841 patches
        model.enable_patch_level_mode()
        x_p = ow.synthetic_images(1, 256, seed=4321)
        o = model(x_p)
        assert (model.batch, model.num_patches) == (1, 841)
        emb = t2n(o["latent_space"])
        rows = np.array([0, 1, 2, 28, 29, 30, 57, 400, 420, 421, 811, 812, 838, 839, 840, 500])
        out["patch_logits"] = t2n(o["classifier"])
        out["patch_rows"] = rows
        out["patch_emb_rows"] = emb[rows]
        out["patch_emb_rowsum"] = emb.astype(np.float64).sum(1)
        out["patch_emb_rownorm"] = np.sqrt((emb.astype(np.float64) ** 2).sum(1))
        # small patch-level case kept in full: 64x48 image -> 5x3 = 15 patches
        x_ps = ow.synthetic_images(2, 64, seed=99)[:, :, :, :48].contiguous()
        o = model(x_ps)
        out["psmall_logits"], out["psmall_emb"] = t2n(o["classifier"]), t2n(o["latent_space"])
        out["psmall_bp"] = np.array([model.batch, model.num_patches])
        model.disable_patch_level_mode()
        # predict_step label logic (models.py:311-333) on a synthetic gt stack
        gts = torch.zeros(3, 1, 8, 8); gts[1, 0, 2, 3] = 1.0
        out["gt2label_bin"] = np.array(rc.gt2label(gts))
        out["gt2label_multi"] = np.array(rc.gt2label(gts, negative=-1, positive=4))
        out["multiclass2binary"] = t2n(rc.multiclass2binary(torch.tensor([0, 1, 2, 3, 0])))
    np.savez_compressed(os.path.join(HERE, "forward.npz"), **out)
    full_patch_emb = emb

    # ---------------- (iii) training step (train-mode BN, autograd) ----------------
    model = rm.PeraNet()
    model.load_state_dict(sd, strict=True)
    model.train()
    model.trainer = types.SimpleNamespace(max_epochs=10)
    xb = ow.synthetic_images(8, 64, seed=55)
    yb = ow.synthetic_labels(8, seed=56)
    loss = model.training_step((xb, yb, None), 0)
    loss.backward()
    tr = {"loss": t2n(loss)}
    names = ["feature_extractor.conv1.weight", "feature_extractor.bn1.weight", "feature_extractor.bn1.bias",
             "feature_extractor.layer1.0.conv1.weight", "feature_extractor.layer2.0.downsample.0.weight",
             "feature_extractor.layer2.0.downsample.1.weight",
             "feature_extractor.layer3.1.conv2.weight", "feature_extractor.layer4.1.bn2.bias",
             "feature_extractor.layer4.1.conv2.weight",
             "concatenator.0.weight", "concatenator.1.weight", "latent_space.0.0.weight", "latent_space.2.1.bias",
             "latent_space.3.weight", "latent_space.3.bias", "latent_space.4.weight", "classifier.weight",
             "classifier.bias"]
    params = dict(model.named_parameters())
    tr["grad_names"] = np.array(names)
    tr["grad_norms"] = np.array([params[n].grad.double().norm().item() for n in names])
    tr["grad_classifier_weight"] = t2n(params["classifier.weight"].grad)
    tr["grad_classifier_bias"] = t2n(params["classifier.bias"].grad)
    tr["grad_bn1_bias"] = t2n(params["feature_extractor.bn1.bias"].grad)
    tr["grad_conv1_slice"] = t2n(params["feature_extractor.conv1.weight"].grad[:4])
    tr["grad_l4_conv2_slice"] = t2n(params["feature_extractor.layer4.1.conv2.weight"].grad[:2, :8])
    bufs = dict(model.named_buffers())
    tr["bn1_running_mean"] = t2n(bufs["feature_extractor.bn1.running_mean"])
    tr["bn1_running_var"] = t2n(bufs["feature_extractor.bn1.running_var"])
    tr["l4_bn2_running_var"] = t2n(bufs["feature_extractor.layer4.1.bn2.running_var"])
    # one SGD step exactly as configure_optimizers builds it (models.py:336-341)
    model.lr, model.num_epochs, model.stage = 0.03, 10, "projection_train"
    (opt,), _ = model.configure_optimizers()
    opt.step()
    tr["post_step_classifier_weight"] = t2n(params["classifier.weight"])
    tr["post_step_conv1_slice"] = t2n(params["feature_extractor.conv1.weight"][:4])
    # second step to exercise the momentum buffer
    opt.zero_grad()
    loss2 = model.training_step((xb, yb, None), 1)
    loss2.backward()
    opt.step()
    tr["loss2"] = t2n(loss2)
    tr["post_step2_classifier_weight"] = t2n(params["classifier.weight"])
    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **tr)

    # ---------------- (iv) AnomalyDetector ----------------
    det = {}
    bank_src = full_patch_emb                                 # 841 patch embeddings of one "good" image
    model = rm.PeraNet(); model.load_state_dict(sd, strict=True); model.eval(); model.enable_patch_level_mode()
    with torch.no_grad():
        q = model(ow.synthetic_images(2, 256, seed=2468))["latent_space"]
    np.random.seed(7)
    d = rm.AnomalyDetector(patch_level=True, batch=2, num_patches=841)
    d.fit(torch.from_numpy(bank_src))
    det["threshold"] = np.float64(d.threshold)
    det["bank_rows"] = np.int64(d.nbrs.n_samples_fit_)
    det["scores"] = t2n(d.predict(q))
    # split-free variant: bank given explicitly, the kernel-level contract
    bank = ow.synthetic_bank(588, 512, seed=2)
    qs = ow.synthetic_bank(300, 512, seed=3)
    det["kernel_scores"] = t2n(osc.sklearn_knn_mean(bank.numpy(), qs.numpy(), 3))
    # image-level (no reshape)
    d2 = rm.AnomalyDetector()
    np.random.seed(11)
    d2.fit(bank)
    det["img_threshold"] = np.float64(d2.threshold)
    det["img_scores"] = t2n(d2.predict(qs[:17]))
    np.savez_compressed(os.path.join(HERE, "detector.npz"), **det)

    # ---------------- (v) upsample: restated third-party blur + torch bilinear ----------------
    g = torch.Generator().manual_seed(5)
    maps = torch.rand(3, 1, 29, 29, generator=g) - 0.2
    onehot = torch.zeros(1, 1, 29, 29); onehot[0, 0, 0, 0] = 1; onehot[0, 0, 14, 20] = 2; onehot[0, 0, 28, 27] = -1
    maps = torch.cat([maps, onehot])
    np.savez_compressed(os.path.join(HERE, "upsample.npz"), maps=t2n(maps),
                        kernel1d=t2n(osc.gaussian_kernel1d(7)),
                        blurred=t2n(osc.gaussian_blur(maps, 7)),
                        up256=t2n(osc.upsample(maps, 256)), up64=t2n(osc.upsample(maps, 64)))

    # ---------------- (vi) cut-paste primitives under a fixed python RNG ----------------
    from PIL import Image
    cp = {}
    base = (ow.synthetic_images(1, 256, seed=31, normalized=False)[0].permute(1, 2, 0) * 255).round().byte().numpy()
    img = Image.fromarray(base, "RGB")
    cp["base"] = base
    random.seed(123)
    patch = rg.generate_patch(img, area_ratio=(0.03, 0.07), aspect_ratio=((0.3, 0.5), (1, 3.3)))
    cp["patch"] = np.array(patch)
    random.seed(124)
    mask = rg.rect2poly(patch, regular=False, sides=8)
    cp["mask_rgba"] = np.array(mask)
    coords = rg.check_valid_coordinates_by_container((256, 256), patch.size, current_coords=(250, 10),
                                                      container_scaling_factor=1.75)
    cp["coords"] = np.array(coords)
    cp["pasted"] = np.array(rg.paste_patch(img, patch, coords, mask))
    c = rg.Container((256, 256), 1.75)
    cp["container"] = np.array([c.center, c.dim, c.left, c.top, c.right, c.bottom, c.width, c.height])
    cp["color_sim"] = np.float64(rg.check_color_similarity(img.crop((0, 0, 40, 40)), patch))
    random.seed(125)
    avg = rg.generate_patch(img, area_ratio=(0.2, 0.5), colorized=True, color_type="average")
    cp["avg_patch_size"] = np.array(avg.size)
    cp["avg_patch_color"] = np.array(avg)[0, 0]
    np.savez_compressed(os.path.join(HERE, "cutpaste.npz"), **cp)

    # ---------------- (vii) AUROC through sklearn (metrics.py:49-56) ----------------
    from sklearn.metrics import roc_curve, auc
    g = torch.Generator().manual_seed(9)
    labels = (torch.rand(4096, generator=g) > 0.8).int().numpy()
    scores = (torch.rand(4096, generator=g) + 0.5 * torch.from_numpy(labels)).numpy().astype(np.float32)
    fpr, tpr, _ = roc_curve(labels, scores)
    np.savez_compressed(os.path.join(HERE, "auroc.npz"), labels=labels, scores=scores, auroc=np.float64(auc(fpr, tpr)))
    make_gradcam(rm, sd)
    make_getitem()
    make_metrics()
    print("fixtures written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"  {f:20s} {os.path.getsize(os.path.join(HERE, f)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
