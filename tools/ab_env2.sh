# like tools/ab_env.sh for any bench arguments: bash tools/ab_env2.sh VAR "bench args"
V=$1; A="$2 --no-cpu-baseline --no-e2e --no-wrn50 --no-faithful --no-precision16 --no-partition-extra --steps 20 --warmup 5"
for r in 1 2; do for v in 0 1; do env $V=$v python bench.py $A > gpurun_out/abe_$v$r.json 2>/dev/null; python -c "import json; print('$V=$v', json.load(open('gpurun_out/abe_$v$r.json'))['train_ms_per_step'])"; done; done
