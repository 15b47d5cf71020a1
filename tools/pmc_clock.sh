#!/bin/bash
# Effective clock of one training-conv op: GRBM_GUI_ACTIVE / 8 / kernel wall time (MI355X_MICROARCH.md, DVFS give-back).
# usage (GPU box): tools/pmc_clock.sh "wgrad 256 32 128 128 3 1 1 20" ...
R=$PWD; LOG=$R/gpurun_out/pmc_clock.log; cd /tmp && export TMPDIR=/tmp
for shape in "$@"; do
  tag=$(echo $shape | tr ' ' '_')
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pc_$tag -o c -- python3 $R/tools/one_wgrad.py $shape > /tmp/pc_$tag.log 2>&1 || { echo "$shape: failed" >> $LOG; tail -3 /tmp/pc_$tag.log >> $LOG; exit 1; }
  python3 - "$tag" >> $LOG <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
dur = {}
for f in glob.glob(f"/tmp/pc_{tag}/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Dispatch_Id"]] = (row["Kernel_Name"], int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
agg = collections.defaultdict(list)
for f in glob.glob(f"/tmp/pc_{tag}/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(dict)
    for row in csv.DictReader(open(f)):
        per[row["Dispatch_Id"]][row["Counter_Name"]] = per[row["Dispatch_Id"]].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    for d, c in per.items():
        if d in dur and ("wgrad_f32" in dur[d][0] or "conv_igemm" in dur[d][0] or "wgrad_bf16" in dur[d][0]):
            name, ns = dur[d]
            agg[name[:70]].append((ns, c.get("GRBM_GUI_ACTIVE", 0) / 8 / ns, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (c.get("GRBM_GUI_ACTIVE", 1) / 8)))
for name, v in agg.items():
    v = v[len(v) // 2:]          # later dispatches: warmed up
    print(f"{tag}: {name}: n={len(v)} dur {sum(x[0] for x in v) / len(v) / 1e3:.1f} us  clock {sum(x[1] for x in v) / len(v):.3f} GHz  mfma_busy/cycles {sum(x[2] for x in v) / len(v):.3f}")
PY
done
