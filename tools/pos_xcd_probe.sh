#!/bin/bash
# Round 4: position order of the position-major convs -- SSAD_POS_LPT=1 (heaviest position first over chunks of 32 sample groups,
# rounds 2-3) against 2 (the same order inside XCD-local blocks of a few sample groups): scoring time per kernel, bit-identity of
# the embeddings, and fabric traffic (FETCH_SIZE / WRITE_SIZE in separate PMC passes, MI355X_MICROARCH.md).
#   tools/pos_xcd_probe.sh [modes...] > gpurun_out/pos_xcd_probe.log
R=$PWD; OUT=$R/gpurun_out
for mode in ${@:-1 2}; do
  echo "== SSAD_POS_LPT=$mode SSAD_POS_XCD_WGS=${SSAD_POS_XCD_WGS:-64}"
  export SSAD_POS_LPT=$mode SSAD_ALLOW_RANDOM_BACKBONE=1
  cd $R && timeout -k 5 300 python - <<'PY' || exit 1
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
    R = 5
    ops.PROFILE = []
    for _ in range(R):
        m(x)
    recs = ops.drain_profile()
ops.PROFILE = None
n = len(recs) // R
tot = {}
for i in range(n):
    ms = sorted(recs[i + k * n]["ms"] for k in range(R))[R // 2]
    if recs[i]["kernel"] == "conv_igemm_pos_f32":
        print(f"   launch {i:3d} {recs[i]['tile']} {ms:7.3f} ms {recs[i]['exec_flops'] / ms / 1e9:6.1f} TF/s")
    key = (recs[i]["kernel"], recs[i].get("tile"))
    e = tot.setdefault(key, [0, 0.0, 0.0]); e[0] += 1; e[1] += ms; e[2] += recs[i]["exec_flops"]
for k, v in tot.items():
    print(f"{k[0]:22s} {str(k[1]):28s} x{v[0]:3d} {v[1]:9.3f} ms  {v[2] / max(v[1], 1e-9) / 1e9:6.1f} TF/s executed")
print("total", round(sum(v[1] for v in tot.values()), 3), "ms for 128 images")
PY
  if [ -z "$NO_PMC" ]; then
    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/prof_${c}_$mode
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_${c}_$mode -o c -- python3 $R/bench.py --phase score --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-faithful --no-partition-extra > /tmp/prof_${c}_$mode.json 2> /tmp/prof_${c}_$mode.err || { tail -5 /tmp/prof_${c}_$mode.err; exit 1; }
    done
    python3 $R/tools/traffic_json.py /tmp/prof_FETCH_SIZE_$mode /tmp/prof_WRITE_SIZE_$mode $OUT/r04_traffic_lpt$mode.json 107648
  fi
done
