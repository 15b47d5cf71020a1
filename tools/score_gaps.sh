#!/bin/bash
# Where the idle time between kernels of a scoring pass is: rocprofv3 --kernel-trace of `bench.py --phase score`, run TWICE on the
# same box (the first process on a fresh box measured 387 ms per pass against 375 in every later one, with identical kernel times).
#   bash tools/score_gaps.sh  -> gpurun_out/score_gaps.log
R=$PWD; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for run in first second; do
  rm -rf /tmp/sg_$run
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sg_$run -o s -- python3 $R/bench.py --phase score --no-cpu-baseline --no-e2e --no-wrn50 --no-partition-extra --no-faithful --steps 5 --warmup 3 > /tmp/sg_$run.json 2>/tmp/sg_$run.err || { tail -5 /tmp/sg_$run.err; exit 1; }
  RUN=$run python3 - <<'PY'
import csv, glob, os, json
run = os.environ['RUN']
f = glob.glob('/tmp/sg_%s/**/*kernel_trace.csv' % run, recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
line = json.loads(open('/tmp/sg_%s.json' % run).read().strip().splitlines()[-1])
# the last 5 passes-pairs: find stem_patch launches (2 per step), take the last 10
idx = [i for i, r in enumerate(rows) if 'stem_patch_fused' in r['Kernel_Name']]
lo = idx[-10]
t0 = int(rows[lo]['Start_Timestamp']); prev_end = t0; busy = 0; gaps = []
for r in rows[lo:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > prev_end: gaps.append(((s - prev_end) / 1e3, r['Kernel_Name'].split('(')[0][-60:]))
    busy += e - s; prev_end = max(prev_end, e)
span = (prev_end - t0) / 1e6
print(run, 'bench line: maps/s', line['anomaly_maps_per_sec'], 'ms/step', line['score_ms_per_step'], '| traced last 5 steps: span %.1f ms, kernels %.1f ms, gaps %.1f ms in %d gaps' % (span, busy / 1e6, sum(g for g, _ in gaps) / 1e3, len(gaps)))
for g, k in sorted(gaps, reverse=True)[:8]: print('    %8.1f us before %s' % (g, k))
PY
done
