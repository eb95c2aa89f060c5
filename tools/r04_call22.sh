#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_spec; mkdir -p $OUT
cd $ROOT
: > $OUT/waves.txt
for w in 1 2 4 8; do
  echo "== PCX_GENERIC_WAVES=$w" >> $OUT/waves.txt
  PCX_GENERIC_WAVES=$w python tools/generic_timing.py 2>&1 | grep pcx_generic >> $OUT/waves.txt
done
echo "== 1,048,576 environments, default" >> $OUT/waves.txt
python tools/generic_timing.py warehouse_L0:1048576 walkers_scroll_groups:1048576 marauders_custom_A:262144 2>&1 | grep pcx_generic >> $OUT/waves.txt
cat $OUT/waves.txt | cut -c1-110
