#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_workers; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_generic_specialised.py -q -m gpu -p no:cacheprovider -x -k "worker or tuner" > $OUT/tests.txt 2>&1; echo "rc=$?"; tail -3 $OUT/tests.txt | cut -c1-200; grep "^E  " $OUT/tests.txt | head -8 | cut -c1-200
: > $OUT/sweep.txt
for combo in "0 0" "2 1" "3 1" "3 2" "4 1" "4 2" "5 2" "6 2" "6 3"; do
  set -- $combo
  echo "== workers $1 lock $2" >> $OUT/sweep.txt
  if [ "$1" == "0" ]; then timeout 200 python tools/generic_timing.py warehouse_L0:262144 walkers_scroll_groups:262144 hello_world:262144 better_scrolly_custom_B:262144 2>&1 | grep pcx_generic | cut -c1-100 >> $OUT/sweep.txt
  else PCX_GENERIC_WORKERS=$1 PCX_GENERIC_LOCK=$2 timeout 200 python tools/generic_timing.py warehouse_L0:262144 walkers_scroll_groups:262144 hello_world:262144 better_scrolly_custom_B:262144 2>&1 | grep pcx_generic | cut -c1-100 >> $OUT/sweep.txt; fi
done
cat $OUT/sweep.txt
