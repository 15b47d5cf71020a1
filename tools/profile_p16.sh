#!/bin/bash
# Precision-16 (half tensors) training step on the GPU box: rocprofv3 kernel stats of the bench command, then the two PMC traffic
# passes (FETCH_SIZE / WRITE_SIZE separately, MI355X_MICROARCH.md) summed over one replayed step.
#   bash tools/profile_p16.sh r05   -> gpurun_out/<tag>_p16_*  (copy the summaries into profiles/)
TAG=${1:-r06}
R=$PWD; OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
ARGS="--phase train --train-precision 16 --no-cpu-baseline --no-e2e --no-wrn50 --no-partition-extra --no-faithful"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p16_stats -o s -- python3 $R/bench.py $ARGS --steps 10 --warmup 3 > $OUT/${TAG}_p16_bench_line_under_rocprof.json 2> /tmp/p16_stats.err || { tail -5 /tmp/p16_stats.err; exit 1; }
cp $(find /tmp/p16_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_p16_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p16_$c -o c -- python3 $R/bench.py $ARGS --steps 3 --warmup 3 > /tmp/p16_$c.json 2> /tmp/p16_$c.err || { tail -5 /tmp/p16_$c.err; exit 1; }
done
python3 $R/tools/p16_traffic_json.py /tmp/p16_FETCH_SIZE /tmp/p16_WRITE_SIZE $OUT/${TAG}_p16_traffic.json 256
