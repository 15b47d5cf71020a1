# scoring pass with the alternative implicit-GEMM tiles for Cout = 128 layers (SSAD_CONV128_VARIANT: 0 = 128 x 128 default, 1 = 256 x 128 eight waves, 2 = 256 x 128 four waves)
for v in 0 2 1; do
  SSAD_CONV128_VARIANT=$v python bench.py --phase score --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-faithful 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v:', d['anomaly_maps_per_sec'], 'maps/s', d['kernel_ms']['score'].get('conv_igemm_pos_f32'))"
done
