R=$PWD; OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-wrn50 --no-partition-extra --no-faithful > $OUT/r04_bench_line_under_rocprof.json 2> /tmp/prof_stats.err || { tail -5 /tmp/prof_stats.err; exit 1; }
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/r04_bench_kernel_stats.csv
