"""Evaluation metrics (mirrors src/self_supervised/metrics.py:49-228): ROC/AUC through scikit-learn as the
reference does, F1 at a threshold, and the MVTec per-region-overlap (PRO) curve with its clipped trapezoid area.
The host functions are the parity statement (pinned to the reference's own metrics.py by tests/golden/metrics.npz); the `*_gpu`
functions below compute the same quantities for device-resident maps with the hand-written radix sort of csrc/auroc.hip
(three argsorts of 83 x 65 536 scores cost ~9 s per category on the host -- more than training that category on this card)."""
from bisect import bisect

import numpy as np
import torch
from scipy import ndimage


def metrics_to_dataframe(metric_dict: dict, objects: list = None):
    """metrics.py:15-20 of the reference: one column per metric, one row per category."""
    import pandas as pd
    if objects is None:
        return pd.DataFrame(metric_dict, columns=metric_dict.keys())
    return pd.DataFrame(metric_dict, columns=metric_dict.keys(), index=objects)


def export_dataframe(dataframe, saving_path: str = None, name: str = 'report.csv', mode='csv') -> None:
    """metrics.py:23-39: csv / latex / markdown export (markdown needs `tabulate`, as in pandas)."""
    import os
    if saving_path and not os.path.exists(saving_path):
        os.makedirs(saving_path)
    target = (saving_path + name) if saving_path else name
    fmt = "%.2f" if (saving_path or mode == 'csv') else "%.3f"
    if mode == 'latex':
        dataframe.to_latex(target, float_format=fmt)
    if mode == 'markdown':
        dataframe.to_markdown(target)
    if mode == 'csv':
        dataframe.to_csv(target, float_format="%.2f")


def _np(a):
    return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)


def compute_roc(labels, scores):
    from sklearn.metrics import roc_curve
    fpr, tpr, thr = roc_curve(_np(labels), _np(scores))
    return np.array(fpr), np.array(tpr), np.array(thr)


def compute_auc(false_positive_rate, true_positive_rate):
    from sklearn.metrics import auc
    return auc(false_positive_rate, true_positive_rate)


def compute_f1(targets, predictions, threshold) -> float:
    """Binary F1 of (predictions >= threshold) against targets > 0."""
    t = _np(targets).ravel() > 0
    p = _np(predictions).ravel() >= float(threshold)
    tp = float(np.sum(p & t))
    denom = 2 * tp + float(np.sum(p & ~t)) + float(np.sum(~p & t))
    return 2 * tp / denom if denom else 0.0


def best_f1_threshold(scores, targets):
    """Threshold maximising F1 over the precision-recall curve (tools.py:141-146 of the reference: torchmetrics'
    PrecisionRecallCurve(), then t[argmax(2PR / (P + R + 1e-10))]).  The curve is restated the way torchmetrics 0.8-0.10
    builds it -- one point per distinct score, counts accumulated and divided in float32, cut at the first full-recall point,
    ascending thresholds -- so that near-ties in F1 resolve as they do there."""
    s = torch.as_tensor(_np(scores)).flatten().float()
    t = (torch.as_tensor(_np(targets)).flatten() == 1).to(torch.long)
    order = torch.argsort(s, descending=True)
    s, t = s[order], t[order]
    idx = torch.cat([torch.where(s[1:] - s[:-1])[0], torch.tensor([t.numel() - 1])])
    tps = torch.cumsum(t * 1.0, dim=0)[idx]
    fps = 1 + idx - tps
    last = int(torch.where(tps == tps[-1])[0][0])
    precision = torch.cat([torch.flip((tps / (tps + fps))[:last + 1], [0]), torch.ones(1)])
    recall = torch.cat([torch.flip((tps / tps[-1])[:last + 1], [0]), torch.zeros(1)])
    thr = torch.flip(s[idx][:last + 1], [0])
    f1 = (2 * precision * recall) / (precision + recall + 1e-10)
    return float(thr[int(np.argmax(f1.numpy()))])


def compute_iou(scores, targets, threshold) -> float:
    """tools.py:131-139: torchmetrics JaccardIndex(2, threshold=) -- mean over the classes {normal, anomalous} of
    intersection / union of (scores >= threshold) against targets > 0; a class that occurs in neither scores 0."""
    t = _np(targets).ravel() > 0
    p = _np(scores).ravel() >= float(threshold)
    ious = []
    for cls in (False, True):
        inter = float(np.sum((p == cls) & (t == cls)))
        union = float(np.sum((p == cls) | (t == cls)))
        ious.append(inter / union if union else 0.0)
    return float(np.mean(ious))


def compute_pro(anomaly_maps, ground_truth_maps):
    """MVTec PRO curve: sweep the threshold down over all pixels; FPR over defect-free pixels, PRO = mean over
    connected ground-truth regions of the covered fraction.  Returns (fprs, pros) from (0,0) to (1,1)."""
    maps = np.asarray(anomaly_maps, dtype=np.float64)
    gts = np.asarray(ground_truth_maps)
    assert maps.size < np.iinfo(np.uint32).max, 'Potential overflow when using np.cumsum(), consider using np.uint64.'
    fp_w = np.zeros(maps.shape, dtype=np.float64)
    pro_w = np.zeros(maps.shape, dtype=np.float64)
    n_regions = 0
    eight = np.ones((3, 3), dtype=int)
    for i, gt in enumerate(gts):
        lab, n = ndimage.label(gt, eight)
        n_regions += n
        fp_w[i][lab == 0] = 1.0
        if n:
            sizes = np.bincount(lab.ravel())[1:]
            inv = np.concatenate([[0.0], 1.0 / sizes])
            pro_w[i] = inv[lab]
    n_ok = fp_w.sum()
    order = np.argsort(maps.ravel(), kind="stable")[::-1]
    s = maps.ravel()[order]
    fprs = (np.cumsum(fp_w.ravel()[order]) / max(n_ok, 1)).astype(np.float32)
    pros = np.cumsum(pro_w.ravel()[order]) / max(n_regions, 1)
    keep = np.append(np.diff(s) != 0, True)            # one point per distinct threshold: the last of each run
    fprs, pros = np.clip(fprs[keep], None, 1.0), np.clip(pros[keep], None, 1.0)
    return np.concatenate(([0.0], fprs, [1.0])), np.concatenate(([0.0], pros, [1.0]))


def trapezoid(x, y, x_max: float = None) -> float:
    """Trapezoid rule on sorted x with an optional upper limit (y interpolated at x_max)."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    ok = np.isfinite(x) & np.isfinite(y)
    if not ok.all():
        print("WARNING: Not all x and y values passed to trapezoid(...) are finite. Will continue with only the finite values.")
    x, y = x[ok], y[ok]
    extra = 0.0
    if x_max is not None:
        if x_max not in x:
            i = bisect(x, x_max)
            assert 0 < i < len(x)
            y_at = y[i - 1] + (y[i] - y[i - 1]) * (x_max - x[i - 1]) / (x[i] - x[i - 1])
            extra = 0.5 * (y_at + y[i - 1]) * (x_max - x[i - 1])
        sel = x <= x_max
        x, y = x[sel], y[sel]
    return float(np.sum(0.5 * (y[1:] + y[:-1]) * (x[1:] - x[:-1])) + extra)


def compute_aupro(all_fprs, all_pros, integration_limit: float) -> float:
    return trapezoid(all_fprs, all_pros, x_max=integration_limit) / integration_limit


def auroc_gpu(labels, scores) -> float:
    """AUROC of GPU-resident scores without leaving the device (csrc/auroc.hip): equals compute_auc(*compute_roc(...))."""
    from . import _hip
    s = scores.detach().reshape(-1).float().contiguous()
    l = (labels.detach().reshape(-1) > 0).to(torch.uint8).to(s.device).contiguous()
    if not s.is_cuda:
        raise RuntimeError("auroc_gpu needs GPU tensors; use compute_roc / compute_auc on the host")
    n = s.numel()
    nbytes = _hip.lib().ssad_auroc_workspace(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=s.device)
    out = torch.empty(2, dtype=torch.float64, device=s.device)
    _hip.check(_hip.lib().ssad_auroc(s.data_ptr(), l.data_ptr(), n, ws.data_ptr(), nbytes, out.data_ptr(), _hip.stream()))
    return float(out[0].item())



def _dev_flat(scores, targets=None):
    s = scores.detach().reshape(-1).float().contiguous()
    if not s.is_cuda:
        raise RuntimeError("the *_gpu metrics need GPU tensors; use the host functions of this module otherwise")
    t = None if targets is None else (torch.as_tensor(targets).detach().reshape(-1).to(s.device) > 0).to(torch.uint8).contiguous()
    return s, t


def best_f1_threshold_gpu(scores, targets) -> float:
    """best_f1_threshold for device-resident scores (csrc/auroc.hip: descending radix sort with the label as payload, running
    count of positives, the curve in torchmetrics' float32 arithmetic, arg-max with its tie rule)."""
    from . import _hip
    s, _ = _dev_flat(scores)
    t = (torch.as_tensor(targets).detach().reshape(-1).to(s.device) == 1).to(torch.uint8).contiguous()
    n = s.numel()
    nbytes = _hip.lib().ssad_best_f1_workspace(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=s.device)
    out = torch.empty(2, dtype=torch.float32, device=s.device)
    _hip.check(_hip.lib().ssad_best_f1_threshold(s.data_ptr(), t.data_ptr(), n, ws.data_ptr(), nbytes, out.data_ptr(), _hip.stream()))
    return float(out[0].item())


def confusion_gpu(scores, targets, threshold):
    """(tp, fp, fn, tn) of (scores >= threshold) against (targets > 0) for device-resident scores."""
    from . import _hip
    s, t = _dev_flat(scores, targets)
    out = torch.zeros(4, dtype=torch.int64, device=s.device)
    _hip.check(_hip.lib().ssad_confusion_counts(s.data_ptr(), t.data_ptr(), s.numel(), float(threshold), out.data_ptr(), _hip.stream()))
    return tuple(int(v) for v in out.tolist())


def compute_f1_gpu(targets, predictions, threshold) -> float:
    tp, fp, fn, _ = confusion_gpu(predictions, targets, threshold)
    denom = 2 * tp + fp + fn
    return 2 * tp / denom if denom else 0.0


def compute_iou_gpu(scores, targets, threshold) -> float:
    tp, fp, fn, tn = confusion_gpu(scores, targets, threshold)
    ious = []
    for inter, union in ((tn, tn + fp + fn), (tp, tp + fp + fn)):          # class "normal", class "anomalous"
        ious.append(inter / union if union else 0.0)
    return float(np.mean(ious))


def compute_pro_gpu(anomaly_maps, ground_truth_maps):
    """compute_pro for device-resident maps [n][H][W]: the ground-truth regions are labelled on the host (a few dozen small masks),
    the 6 M scores are sorted, weighted, accumulated and compacted on the device (ssad_pro_curve).  Returns (fprs, pros) as numpy
    arrays from (0, 0) to (1, 1); equal to compute_pro up to the summation order of the fp64 running sum (~1e-13)."""
    from . import _hip
    maps = anomaly_maps.detach().float().contiguous()
    if not maps.is_cuda:
        raise RuntimeError("compute_pro_gpu needs GPU maps; use compute_pro on the host")
    gts = _np(ground_truth_maps)
    gts = gts.reshape((-1,) + gts.shape[-2:])
    fp_w = np.zeros(gts.shape, dtype=np.uint8)
    pro_w = np.zeros(gts.shape, dtype=np.float64)
    n_regions = 0
    eight = np.ones((3, 3), dtype=int)
    for i, gt in enumerate(gts):
        lab, k = ndimage.label(gt, eight)
        n_regions += k
        fp_w[i] = lab == 0
        if k:
            sizes = np.bincount(lab.ravel())[1:]
            pro_w[i] = np.concatenate([[0.0], 1.0 / sizes])[lab]
    n = maps.numel()
    assert n == fp_w.size, "maps and ground truths differ in size"
    n_ok = float(fp_w.sum())
    dev = maps.device
    fpd, prd = torch.from_numpy(fp_w.reshape(-1)).to(dev), torch.from_numpy(pro_w.reshape(-1)).to(dev)
    nbytes = _hip.lib().ssad_pro_curve_workspace(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    fprs = torch.empty(n, dtype=torch.float32, device=dev)
    pros = torch.empty(n, dtype=torch.float64, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    _hip.check(_hip.lib().ssad_pro_curve(maps.data_ptr(), fpd.data_ptr(), prd.data_ptr(), n, max(n_ok, 1.0), float(max(n_regions, 1)),
                                         ws.data_ptr(), nbytes, fprs.data_ptr(), pros.data_ptr(), count.data_ptr(), _hip.stream()))
    k = int(count.item())
    f, p = fprs[:k].cpu().numpy(), pros[:k].cpu().numpy()
    return np.concatenate(([0.0], f, [1.0])), np.concatenate(([0.0], p, [1.0]))
