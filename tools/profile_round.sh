#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the bench command, PMC traffic passes of the scoring phase (FETCH_SIZE /
# WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) and SQ counters of the training kernels.
#   tools/profile_round.sh r04      -> gpurun_out/<tag>_*  (copy the summaries into profiles/)
# rocprofv3 wraps python3 itself (never a launcher); --gpus 1 only.
TAG=${1:-r06}
R=$PWD; OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
set -x
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-wrn50 --no-partition-extra --no-faithful --no-precision16 > $OUT/${TAG}_bench_line_under_rocprof.json 2> /tmp/prof_stats.err || { tail -5 /tmp/prof_stats.err; exit 1; }
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o c -- python3 $R/bench.py --phase score --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-faithful --no-precision16 --no-partition-extra > /tmp/prof_$c.json 2> /tmp/prof_$c.err || { tail -5 /tmp/prof_$c.err; exit 1; }
done
python3 $R/tools/traffic_json.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE $OUT/${TAG}_traffic.json 107648 /tmp/prof_FETCH_SIZE.json
rm -f $OUT/pmc_op.log
cd $R && bash $R/tools/pmc_op.sh "c64 256 64 64 64" "wgradh 256 16 256 256" "igemm 256 32 128 128" "igemm 256 16 256 256" "dgrad 256 16 256 256" && cp $OUT/pmc_op.log $OUT/${TAG}_pmc_training_kernels.txt
