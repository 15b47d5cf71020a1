#!/bin/bash
# PMC passes for one training-conv op: tools/pmc_wgrad.sh "wgrad 256 32 128 128 3 1 1" ...   (run on the GPU box)
R=$PWD; LOG=$R/gpurun_out/pmc_wgrad.log; cd /tmp && export TMPDIR=/tmp
for shape in "$@"; do
  tag=$(echo $shape | tr ' ' '_')
  echo "== $shape" >> $LOG
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc ${PMC1:-SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES} --output-format csv -d /tmp/pm1_$tag -o a -- python3 $R/tools/one_wgrad.py $shape > /tmp/pm1_$tag.log 2>&1 || { echo "pass 1 failed/timeout" >> $LOG; tail -3 /tmp/pm1_$tag.log >> $LOG; exit 1; }
  echo "pass 1 done" >> $LOG
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc ${PMC2:-FETCH_SIZE} --output-format csv -d /tmp/pm2_$tag -o b -- python3 $R/tools/one_wgrad.py $shape > /tmp/pm2_$tag.log 2>&1 || { echo "pass 2 failed/timeout" >> $LOG; tail -3 /tmp/pm2_$tag.log >> $LOG; exit 1; }
  echo "pass 2 done" >> $LOG
  python3 - "$tag" >> $LOG <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
for d in ("pm1_", "pm2_"):
    for f in glob.glob(f"/tmp/{d}{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        seen = set()
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"][:60]
            acc[kn][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (kn, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key); cnt[kn] += 1
        for kn, c in acc.items():
            if "wgrad" in kn or "conv_igemm" in kn:
                print(tag, kn, cnt[kn], {k: f"{v / cnt[kn]:.4g}" for k, v in c.items()})
PY
done
