#!/bin/bash
# PMC study of one training op: clock, MFMA-busy share, wave wait / issue-stall / active shares, LDS conflicts.
# usage (GPU box): tools/pmc_op.sh "c64 256 64 64 64" "wgradh 256 16 256 256" ...   -> gpurun_out/pmc_op.log
R=$PWD; LOG=$R/gpurun_out/pmc_op.log; cd /tmp && export TMPDIR=/tmp
for shape in "$@"; do
  tag=$(echo $shape | tr ' ' '_')
  for pass in 1 2; do
    if [ $pass = 1 ]; then CTR="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; else CTR="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; fi
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d /tmp/po_${tag}_$pass -o c -- python3 $R/tools/one_op.py $shape > /tmp/po_$tag.log 2>&1 || { echo "$shape: failed" >> $LOG; tail -3 /tmp/po_$tag.log >> $LOG; exit 1; }
  done
  python3 - "$tag" >> $LOG <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for pas in (1, 2):
    dur = {}
    for f in glob.glob(f"/tmp/po_{tag}_{pas}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Dispatch_Id"]] = (row["Kernel_Name"], int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for f in glob.glob(f"/tmp/po_{tag}_{pas}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(dict)
        for row in csv.DictReader(open(f)):
            per[row["Dispatch_Id"]][row["Counter_Name"]] = per[row["Dispatch_Id"]].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        for d, c in per.items():
            if d in dur and any(k in dur[d][0] for k in ("conv3x3_c64", "wgrad3x3_halo", "wgrad_f32", "conv_igemm")):
                name, ns = dur[d]
                c = dict(c); c["ns"] = ns
                for k, v in c.items():
                    res[name[:60]][f"{pas}:{k}"].append(v)
for name, c in res.items():
    m = {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in c.items()}
    cyc = m.get("1:GRBM_GUI_ACTIVE", 0) / 8
    out = [f"{tag}: {name}: {m.get('1:ns', 0) / 1e3:.1f} us clock {cyc / max(m.get('1:ns', 1), 1):.3f} GHz",
           f"mfma_busy {m.get('1:SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(cyc, 1):.3f}"]
    wc = m.get("2:SQ_WAVE_CYCLES", 0)
    if wc:
        out.append(f"wave: wait {m.get('2:SQ_WAIT_ANY', 0) / wc:.3f} issue-stall {m.get('2:SQ_WAIT_INST_ANY', 0) / wc:.3f} (lds {m.get('2:SQ_WAIT_INST_LDS', 0) / wc:.3f}) active {m.get('2:SQ_ACTIVE_INST_ANY', 0) / wc:.3f}")
        out.append(f"lds conflict/active {m.get('2:SQ_LDS_BANK_CONFLICT', 0) / max(m.get('2:SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
    print("  ".join(out))
PY
done
