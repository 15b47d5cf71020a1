#!/bin/bash
# PMC passes of one scoring pass (FETCH_SIZE, WRITE_SIZE separately, as MI355X_MICROARCH.md prescribes) and the per-launch-shape
# breakdown of tools/traffic_by_layer.py -> gpurun_out/traffic_by_layer.log
R=$PWD; OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tbl_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tbl_$c -o c -- python3 $R/bench.py --phase score --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-faithful --no-partition-extra > /tmp/tbl_$c.json 2> /tmp/tbl_$c.err || { tail -5 /tmp/tbl_$c.err; exit 1; }
done
python3 $R/tools/traffic_by_layer.py /tmp/tbl_FETCH_SIZE /tmp/tbl_WRITE_SIZE > $OUT/traffic_by_layer.log 2>&1; cat $OUT/traffic_by_layer.log
