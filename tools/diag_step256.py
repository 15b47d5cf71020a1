#!/usr/bin/env python3
"""Diagnostic: per-parameter gradient error of one training step at (B,3,256,256) vs torch-CPU fp32 and fp64 autograd."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-supervised-anomaly-detection_amd"))
os.environ.setdefault("SSAD_ALLOW_RANDOM_BACKBONE", "1")
import torch
import torch.nn.functional as F
from oracle import weights as ow
from oracle.peranet import OraclePeraNet, train_step
from self_supervised import training, ops
from self_supervised.models import PeraNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
do64 = len(sys.argv) > 2
dev = torch.device("cuda:0")
torch.set_num_threads(16)
sd = ow.seeded_state_dict(0)
x, y = ow.synthetic_images(B, 256, seed=1234), ow.synthetic_labels(B, seed=1235)
ref = OraclePeraNet(); ref.load_state_dict(sd); ref.train()
loss, _, out = train_step(ref, x, y); loss.backward()
m = PeraNet(); m.load_state_dict(sd); m.to(dev).train(); m.unfreeze()
st = training.DataParallelStep(m, lr=0.005, world_size=1, graph=False)
logits, emb = st.eng.forward(x.to(dev))
dl = torch.empty_like(logits)
la = ops.softmax_ce(logits, y.to(dev), dl, 1.0 / B)
st.eng.backward(dl)
print("loss", la[0].item(), loss.item())
rp = dict(ref.named_parameters())
gmax = max(p.grad.abs().max().item() for p in ref.parameters())
p64 = None
if do64:
    r64 = copy.deepcopy(ref).double(); r64.zero_grad()
    l64, _, _ = train_step(r64, x.double(), y); l64.backward()
    p64 = dict(r64.named_parameters())
rows = []
for n, p in m.named_parameters():
    g, r = p.grad.detach().cpu().double(), rp[n].grad.double()
    e = (g - r).abs().max().item() / max(r.abs().max().item(), 1e-4 * gmax)
    extra = ""
    if p64 is not None:
        t = p64[n].grad
        extra = f"  |hip-f64| {(g - t).abs().max().item():.3e}  |t32-f64| {(r - t).abs().max().item():.3e}"
    rows.append((e, n, r.abs().max().item(), extra))
for e, n, mx, extra in sorted(rows, reverse=True)[:25]:
    print(f"{e:.3e}  {n:50s} max|g| {mx:.3e}{extra}")
