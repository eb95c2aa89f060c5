#!/bin/bash
# CU-wide streaming slots (pcx_stream.h cu_slot_*) in pcx_warehouse_step's single-wave shape: parity, then a sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_slots; mkdir -p $OUT
cd $ROOT
PCX_CU_SLOTS=2 PCX_WAVES_PER_CU=10 PCX_COOP_BELOW=0 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_random_levels.py -q -m gpu -p no:cacheprovider -k "warehouse" 2>&1 | tail -2
: > $OUT/sweep.txt
for fx in warehouse_L0:262144 warehouse_L0:1048576; do
  for combo in "4 0" "6 0" "8 0" "8 2" "8 3" "8 4" "10 3" "12 2" "12 3" "12 4" "16 3" "16 4"; do
    set -- $combo
    r=$(PCX_FORCE_GENERIC=0 PCX_WAVES_PER_CU=$1 PCX_CU_SLOTS=$2 timeout 120 python tools/generic_timing.py $fx 2>&1 | grep pcx_ | cut -c1-100)
    echo "waves/CU $1 slots $2: $r" >> $OUT/sweep.txt
  done
done
cat $OUT/sweep.txt
