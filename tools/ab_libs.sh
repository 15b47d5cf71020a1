# Same-box A / B of two builds of lib/libssad_hip.so on the precision-16 step (box-to-box spread is 1-2 %, more than most kernel changes):
#   build the old tree's library into ab_tmp/lib_old.so and the new one into ab_tmp/lib_new.so (ab_tmp/ travels with the snapshot), then
#   gpurun -- 'bash tools/ab_libs.sh'      -> two interleaved runs of each, ms per step
L="self-supervised-anomaly-detection_amd/lib/libssad_hip.so"
A="${BENCH_ARGS:---phase train --train-precision 16 --no-cpu-baseline --no-e2e --no-wrn50 --no-faithful --no-precision16 --no-partition-extra --steps 30 --warmup 5}"
for r in 1 2; do for v in old new; do cp ab_tmp/lib_$v.so $L; python bench.py $A > gpurun_out/ab_$v$r.json 2>/dev/null; python -c "import json; print('$v', json.load(open('gpurun_out/ab_$v$r.json')).get('${KEY:-train_ms_per_step}'))"; done; done
cp ab_tmp/lib_new.so $L
