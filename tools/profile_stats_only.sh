#!/bin/bash
# rocprofv3 kernel stats of the headline bench command WITHOUT its side measurements (wrn50, batch-32 partition, bf16x6, end-to-end,
# CPU baseline): every position-major launch in the trace is then a full-size launch of the timed region or its warm-up, so the
# average duration of that kernel can be compared with the line's roofline.avg_launch_ms.  -> gpurun_out/r05_bench_kernel_stats.csv
R=$PWD; OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-wrn50 --no-partition-extra --no-faithful --no-precision16 > $OUT/r05_bench_line_under_rocprof.json 2> /tmp/prof_stats.err || { tail -5 /tmp/prof_stats.err; exit 1; }
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/r05_bench_kernel_stats.csv
