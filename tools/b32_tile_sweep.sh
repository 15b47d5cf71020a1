for sg in 0 260 500 1100; do for tg in 0 200 520; do
  r=$(SSAD_CONV_SMALL_GRID=$sg SSAD_CONV_TINY_GRID=$tg python bench.py --batch 32 --scaling weak --phase train --no-cpu-baseline --no-e2e --no-partition-extra --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['train_ms_per_step'])")
  echo "small=$sg tiny=$tg -> $r"
done; done
