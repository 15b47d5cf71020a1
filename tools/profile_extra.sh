TAG=${1:-r04}
R=$PWD; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_wrn -o s -- python3 $R/bench.py --config wrn50 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_wrn50_bench_line_under_rocprof.json 2>/tmp/p_wrn.err && cp $(find /tmp/p_wrn -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_wrn50_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_b32 -o s -- python3 $R/bench.py --batch 32 --scaling weak --phase train --no-cpu-baseline --no-e2e --no-partition-extra --steps 30 --warmup 5 > $OUT/${TAG}_batch32_bench_line_under_rocprof.json 2>/tmp/p_b32.err && cp $(find /tmp/p_b32 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_batch32_kernel_stats.csv
ls -la $OUT/${TAG}_*
