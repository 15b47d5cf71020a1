R=$PWD; cd /tmp; export TMPDIR=/tmp
for ch in 8 16; do
  export SSAD_POS_CHUNK=$ch
  python3 $R/bench.py --phase score --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $ch maps/s', d['value'])"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pc_$c; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pc_$c -o c -- python3 $R/bench.py --phase score --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2>&1
  done
  python3 $R/tools/traffic_json.py /tmp/pc_FETCH_SIZE /tmp/pc_WRITE_SIZE /tmp/t_$ch.json 107648 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('chunk $ch traffic MB', d['traffic_MB_per_launch'], 'fetch', d['fetch_MB_per_launch'])"
done
unset SSAD_POS_CHUNK
python3 $R/bench.py --phase score --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk 32 maps/s', d['value'])"
