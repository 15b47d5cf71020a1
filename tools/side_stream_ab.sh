# A/B of SSAD_WGRAD_STREAM (weight gradients on a second stream inside the replayed graph) on one box:
#   bash tools/side_stream_ab.sh  -> gpurun_out/side_ab_<batch>_<precision>_<0|1>.json
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
COMMON="--scaling weak --phase train --no-cpu-baseline --no-e2e --no-wrn50 --no-faithful --no-precision16 --no-partition-extra --steps 20 --warmup 5"
for B in 32 256; do for P in 32 16; do for S in 0 1; do
  SSAD_WGRAD_STREAM=$S timeout -k 10 200 python3 bench.py --batch $B --train-precision $P $COMMON > $OUT/side_ab_${B}_${P}_$S.json 2>$OUT/side_ab_${B}_${P}_$S.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$OUT/side_ab_${B}_${P}_$S.json')); print('batch $B precision $P side $S:', d.get('train_ms_per_step'), d.get('self_check',{}).get('launch_mode'), d.get('self_check',{}).get('graph_equals_eager'))"
done; done; done
