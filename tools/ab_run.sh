#!/bin/bash
# Same-box A/B of two builds of libpcx.so: alternates them three times per config.
#   tools/ab_run.sh gpurun_variants/libpcx_noepi.so "marauders 32768" "warehouse 262144"
ALT=$1; shift
for cfg in "$@"; do
  set -- $cfg
  for rep in 1 2 3; do
    for lib in default $ALT; do
      if [ $lib = default ]; then unset PCX_LIB; else export PCX_LIB=$PWD/$lib; fi
      python bench.py --game $1 --batch $2 --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-12s %-8s %-40s %.5f ms  frac %.3f' % ('$1', '$2', '$lib', d['roofline']['kernel_ms'], d['roofline']['frac']))"
    done
  done
done
