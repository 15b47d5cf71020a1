#!/bin/bash
# Round 4: the 256 x 256 position-major tile (SSAD_CONV128_VARIANT=6) against the shipped 128 x 256 / 128 x 128 tiles:
# bit-identity of the scoring embeddings (one process per variant: the switch is read once) and per-launch times.
#   tools/micro/tile256_probe.sh > gpurun_out/tile256_probe.log
set -e
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/micro/mfma_shape.hip -o /tmp/mfma_shape && timeout -k 5 120 /tmp/mfma_shape
for v in 0 6; do
  echo "== SSAD_CONV128_VARIANT=$v"
  SSAD_CONV128_VARIANT=$v SSAD_ALLOW_RANDOM_BACKBONE=1 timeout -k 5 300 python - <<'PY'
import os, sys, hashlib
sys.path.insert(0, "self-supervised-anomaly-detection_amd"); sys.path.insert(0, ".")
import torch
from self_supervised import ops
from self_supervised.models import PeraNet
from oracle import weights as ow
dev = torch.device("cuda:0")
m = PeraNet(); m.load_state_dict(ow.seeded_state_dict(0)); m.to(dev).eval(); m.enable_patch_level_mode()
x = ow.synthetic_images(128, 256, seed=9).to(dev)
with torch.no_grad():
    for _ in range(2):
        out = m(x)
    torch.cuda.synchronize()
    print("embedding sha1", hashlib.sha1(out["latent_space"].cpu().numpy().tobytes()).hexdigest())
    R = 3
    ops.PROFILE = []
    for _ in range(R):
        m(x)
    recs = ops.drain_profile()
ops.PROFILE = None
n = len(recs) // R
tot = {}
for i in range(n):
    ms = sorted(recs[i + k * n]["ms"] for k in range(R))[R // 2]
    key = (recs[i]["kernel"], recs[i].get("tile"))
    e = tot.setdefault(key, [0, 0.0, 0.0]); e[0] += 1; e[1] += ms; e[2] += recs[i]["exec_flops"]
for k, v in tot.items():
    print(f"{k[0]:22s} {str(k[1]):28s} x{v[0]:3d} {v[1]:9.3f} ms  {v[2] / max(v[1], 1e-9) / 1e9:6.1f} TF/s executed")
print("total", round(sum(v[1] for v in tot.values()), 3), "ms for 128 images")
PY
done
