#!/bin/bash
run() { env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-60s %.5f ms  frac %.3f' % ('$*', d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for rep in 1 2 3; do
run PCX_SM_CODES=1
run PCX_SM_CODES=0
for w in 3 4 5; do run PCX_WAVES_PER_WG=2 PCX_WGS_PER_CU=$w; done
done
