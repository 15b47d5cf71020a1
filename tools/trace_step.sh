# per-dispatch kernel trace of one replayed training step: bash tools/trace_step.sh [batch, default 32]
# (--no-faithful --no-precision16: the bf16x6 side measurement would otherwise supply the shortest step of the run)
# BENCH_ARGS="--train-precision 16" TAG=r04_p16 bash tools/trace_step.sh 256   -> the precision-16 step
# -> gpurun_out/<TAG, default r04>_b<batch>_trace_step.csv
B=${1:-32}
R=$PWD; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/t_b$B -o s -- python3 $R/bench.py --batch $B --scaling weak --phase train --no-cpu-baseline --no-e2e --no-wrn50 --no-faithful --no-precision16 --no-partition-extra --steps 6 --warmup 3 $BENCH_ARGS > $OUT/${TAG:-r06}_b${B}_trace_line.json 2>/tmp/t_b$B.err && B=$B python3 - <<'PY'
import csv, glob, os
f = glob.glob('/tmp/t_b%s/**/*kernel_trace.csv' % os.environ['B'], recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step starts at each stem_conv7x7 launch
idx = [i for i, r in enumerate(rows) if 'stem_conv7x7' in r['Kernel_Name']]
spans = [(int(rows[idx[k+1]]['Start_Timestamp']) - int(rows[idx[k]]['Start_Timestamp']), k) for k in range(len(idx) - 1)]
print('step spans (us):', [round(a / 1e3) for a, _ in spans])
k = min(spans)[1]            # a replayed step (eager steps are the long ones)
lo, hi = idx[k], idx[k + 1]
out = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/%s_b%s_trace_step.csv' % (os.environ.get('TAG', 'r04'), os.environ['B'])
with open(out, 'w') as o:
    o.write('kernel,grid,wg,start_us,dur_us,gap_before_us\n')
    t0 = int(rows[lo]['Start_Timestamp']); prev_end = t0
    for r in rows[lo:hi]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        name = name.split('(')[0][:70]
        o.write(f"{name},{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']},{r['Workgroup_Size_X']},{(s-t0)/1e3:.1f},{(e-s)/1e3:.1f},{(s-prev_end)/1e3:.1f}\n")
        prev_end = e
PY
