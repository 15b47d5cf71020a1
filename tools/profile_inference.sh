# rocprofv3 kernel statistics of tools.inference on a fake MVTec category -> gpurun_out/r03_inference_kernel_stats.csv
R=$PWD; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_inf -o s -- python3 $R/tools/profile_inference.py > $OUT/r03_inference_profile.log 2>&1 && cp $(find /tmp/p_inf -name "*kernel_stats.csv" | head -1) $OUT/r03_inference_kernel_stats.csv
