#!/usr/bin/env python3
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from PIL import Image, ImageDraw
from self_supervised import augment, _hip, pil_exact as px
dev = torch.device("cuda:0")
img = np.zeros((128, 128, 3), np.uint8)
aug = augment.GpuCutPaste("carpet", img[None], np.ones((1, 128, 128), bool), cuts_u8=img[None], device=dev)
base = np.zeros((), augment.AUG_DTYPE); base["cut_w"], base["cut_h"] = 128, 128
def run(r):
    params = torch.from_numpy(np.stack([r]).view(np.uint8).reshape(1, -1)).to(dev)
    work = torch.empty((1, 128, 128, 3), dtype=torch.uint8, device=dev)
    gm = torch.empty(1, device=dev); out = torch.empty((1, 3, 128, 128), device=dev)
    _hip.check(_hip.lib().ssad_cutpaste_augment(aug.images.data_ptr(), aug.cuts.data_ptr(), params.data_ptr(), work.data_ptr(),
                                                gm.data_ptr(), out.data_ptr(), 1, 128, 128, 128, 128, aug._mean, aug._std, _hip.stream()))
    return work.cpu()[0].numpy()
rng = random.Random(11)
nbad = 0
for it in range(300):
    width = 3
    npts = rng.choice([2, 3, 6, 31, 32])
    pts = [(rng.uniform(-8, 136), rng.uniform(-8, 136)) for _ in range(npts)]
    ip = px.line_points_int(pts)
    r = base.copy()
    r["label"], r["line_n"], r["line_rgb"], r["line_width"] = 3, npts, (192, 192, 192), width
    r["line_xy"][:2 * npts] = np.asarray(ip, np.int32).ravel()
    for k, ((x0, y0), (x1, y1)) in enumerate(zip(ip[:-1], ip[1:])):
        q = px.wide_line_quad(x0, y0, x1, y1, width)
        if q is not None:
            r["line_quad_ok"][k] = 1; r["line_quad"][8 * k:8 * k + 8] = np.asarray(q, np.int32).ravel()
    got = run(r)[..., 0] > 0
    want = px.draw_line(pts, 128, 128, width)
    if not np.array_equal(got, want):
        nbad += 1
        d = np.argwhere(got != want)
        print("case", it, "npts", npts, "diff", len(d), d[:6].tolist(), [(bool(got[a, b]), bool(want[a, b])) for a, b in d[:6]])
        # which segment is responsible?
        for k, ((x0, y0), (x1, y1)) in enumerate(zip(ip[:-1], ip[1:])):
            q = px.wide_line_quad(x0, y0, x1, y1, width)
            if q is None: continue
            m = px.polygon_fill([v[0] for v in q], [v[1] for v in q], 128, 128)
            if any(m[a, b] for a, b in d[:6]):
                print("   segment", k, "quad", q)
                break
        if nbad > 3: break
print("bad", nbad)
