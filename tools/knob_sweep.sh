for W in 8 12 16; do for D in 0 1 2 3 6; do
  echo -n "waves/CU=$W debug=$D: "; PCX_WAVES_PER_CU=$W PCX_DEBUG=$D python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"
done; done
