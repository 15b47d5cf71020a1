#!/bin/bash
# Weight gradients on a second stream (SSAD_WGRAD_STREAM=1) under hipGraph replay: fork/join become graph edges, no host cost.
# Round 2 measured it eagerly (slower); batch 32 has one workgroup per CU in most conv kernels, so a co-resident wgrad could fill
# the dgrad kernel's latency.  Prints train_ms_per_step at batch 32 and 256, default and with the second stream.
set -e
mkdir -p gpurun_out
F="--phase train --no-cpu-baseline --no-e2e --no-wrn50 --no-partition-extra --no-faithful --steps 30 --warmup 6"
for b in 32 256; do
  for s in 0 1; do
    SSAD_WGRAD_STREAM=$s python bench.py $F --batch $b --global-batch $b > gpurun_out/ws_${b}_${s}.json 2> gpurun_out/ws_${b}_${s}.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/ws_${b}_${s}.json").read().strip().splitlines()[-1])
print("batch", $b, "SSAD_WGRAD_STREAM", $s, "train_ms_per_step", d.get("train_ms_per_step"), d.get("config",{}).get("train_step_launch"))
PY
  done
done
