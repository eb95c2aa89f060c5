#!/bin/bash
# Launch-shape sweep of the round-2 kernels (same box, same run).
run() { # label, env assignments..., -- game batch
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --game $1 --batch $2 --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %-8s %-28s %.5f ms  frac %.3f' % ('$1', '$2', '$label', d['roofline']['kernel_ms'], d['roofline']['frac']))"
}
for w in 4 6 8 10 12 16; do run "waves_per_cu=$w" PCX_WAVES_PER_CU=$w -- warehouse 262144; done
for w in 4 6 8 10 12 16; do run "waves_per_cu=$w" PCX_WAVES_PER_CU=$w -- warehouse 1048576; done
for w in 1 4 8; do run "em_waves=$w" PCX_EM_WAVES=$w -- marauders 32768; done
for w in 4 6 8 12; do run "em_waves=1 waves_per_cu=$w" PCX_EM_WAVES=1 PCX_WAVES_PER_CU=$w -- marauders 262144; done
for w in 4 8; do run "em_waves=$w" PCX_EM_WAVES=$w -- marauders 262144; done
run "coop" PCX_COOP_BELOW=100000 -- warehouse 262144
