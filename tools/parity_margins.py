#!/usr/bin/env python3
"""Max errors of the exact-fp32 and the bf16x3 paths against the reference's vectors (tests/golden): margins of the 1e-4 bar."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
import numpy as np, torch
from oracle import weights as ow
from self_supervised.models import AnomalyDetector, PeraNet
dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests/golden/forward.npz")); gd = np.load(os.path.join(ROOT, "tests/golden/detector.npz"))
sd = ow.seeded_state_dict(0)
for mode in ("f32", "bf16x3"):
    os.environ["SSAD_MATH"] = mode
    m = PeraNet(); m.load_state_dict(sd); m.eval().to(dev)
    with torch.no_grad():
        o = m(ow.synthetic_images(2, 256, seed=1234).to(dev))
        e_log = np.abs(o["classifier"].cpu().numpy() - g["img_logits"]).max()
        e_emb = np.abs(o["latent_space"].cpu().numpy() - g["img_emb"]).max() / max(1.0, np.abs(g["img_emb"]).max())
        m.enable_patch_level_mode()
        bank_src = m(ow.synthetic_images(1, 256, seed=4321).to(dev))["latent_space"]
        q = m(ow.synthetic_images(2, 256, seed=2468).to(dev))["latent_space"]
    e_pe = np.abs(bank_src.cpu().numpy()[g["patch_rows"]] - g["patch_emb_rows"]).max() / max(1.0, np.abs(g["patch_emb_rows"]).max())
    np.random.seed(7)
    d = AnomalyDetector(patch_level=True, batch=2, num_patches=m.num_patches); d.fit(bank_src.cpu())
    e_map = np.abs(d.predict(q).cpu().numpy() - gd["scores"]).max()
    print(f"{mode:7s}: image logits {e_log:.2e}  image emb {e_emb:.2e}  patch emb {e_pe:.2e}  anomaly maps {e_map:.2e} (abs; maps range {gd['scores'].min():.3f}..{gd['scores'].max():.3f})")
