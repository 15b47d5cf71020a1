#!/bin/bash
# A / B of an environment switch on the training phase of bench.py, same box, alternating runs:  bash tools/ab_aside.sh SSAD_ASIDE [runs]
VAR=${1:-SSAD_ASIDE}; RUNS=${2:-2}
ARGS="--phase train --no-cpu-baseline --no-e2e --no-wrn50 --no-faithful --steps 20 --warmup 3"
for i in $(seq 1 $RUNS); do
  for v in ${VALS:-0 1}; do
    env $VAR=$v timeout -k 10 200 python bench.py $ARGS > gpurun_out/ab_${VAR}_${v}_$i.json 2> gpurun_out/ab_${VAR}_${v}_$i.err || exit 1
    python3 - <<PY
import json
d = json.loads(open("gpurun_out/ab_${VAR}_${v}_$i.json").read().strip().splitlines()[-1])
print("$VAR=$v run $i: fp32 %.3f ms  b32 %.3f ms  p16 %.3f ms" % (d["train_ms_per_step"], d["batch32"]["train_ms_per_step"], d["precision16"]["train_ms_per_step"]))
PY
  done
done
