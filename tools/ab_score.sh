#!/bin/bash
# A / B of an environment switch on the scoring phase of bench.py, same box, alternating runs:  bash tools/ab_score.sh SSAD_DEDUP [runs]
VAR=${1:-SSAD_DEDUP}; RUNS=${2:-2}
ARGS="--phase score --no-cpu-baseline --no-e2e --no-wrn50 --no-faithful --no-partition-extra --steps 5 --warmup 2"
for i in $(seq 1 $RUNS); do
  for v in ${VALS:-0 1}; do
    env $VAR=$v timeout -k 10 200 python bench.py $ARGS > gpurun_out/abs_${VAR}_${v}_$i.json 2> gpurun_out/abs_${VAR}_${v}_$i.err || exit 1
    python3 - <<PY
import json
d = json.loads(open("gpurun_out/abs_${VAR}_${v}_$i.json").read().strip().splitlines()[-1])
km = d["kernel_ms"]["score"]
print("$VAR=$v run $i: %.2f maps/s  %.2f ms per 256 images | " % (d["anomaly_maps_per_sec"], d["score_ms_per_step"]) +
      "  ".join("%s %.1f x%d" % (k, v[0], v[1]) for k, v in sorted(km.items(), key=lambda kv: -kv[1][0])[:7]))
PY
  done
done
