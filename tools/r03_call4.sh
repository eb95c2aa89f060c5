#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_call4
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|FAILED" $OUT/suite.log | tail -8
for c in warehouse_L0:262144 walkers_scroll_groups:262144 marauders:32768 hello_world:262144 directives_z_order:262144; do
  echo "== $c"; PCX_DEBUG=8 timeout 120 python tools/generic_timing.py $c 2>&1 | grep -E "cycles per group|pcx_generic" | tail -3
done > $OUT/generic_cycles.txt 2>&1
cat $OUT/generic_cycles.txt
