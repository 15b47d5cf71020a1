# Same-box A / B of an environment switch on the training step: bash tools/ab_env.sh SSAD_CONV32W "--train-precision 32" [batch]
V=$1; P=${2:---train-precision 32}; B=${3:-256}
A="--batch $B --scaling weak --phase train $P --no-cpu-baseline --no-e2e --no-wrn50 --no-faithful --no-precision16 --no-partition-extra --steps 20 --warmup 5"
for r in 1 2; do for v in 0 1; do env $V=$v python bench.py $A > gpurun_out/abe_$v$r.json 2>/dev/null; python -c "import json; print('$V=$v batch $B', json.load(open('gpurun_out/abe_$v$r.json'))['train_ms_per_step'])"; done; done
