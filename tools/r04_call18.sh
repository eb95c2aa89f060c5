#!/bin/bash
# specialised builds of pcx_generic_step (one library per fixture, PCX_LIB) against the table-driven build: parity digests and timing
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_spec; mkdir -p $OUT
cd $ROOT
S=pycolab_amd/csrc/build/spec
: > $OUT/ab.txt
for fx in warehouse_L0:262144 marauders_custom_A:32768 marauders:32768 walkers_scroll_groups:262144 directives_z_order:262144 walkers_room:262144 hello_world:262144 warehouse_L0_unoccluded:262144 better_scrolly_custom_B:262144 marauders_unoccluded:32768; do
  name=${fx%%:*}
  a=$(python tools/experiments/spec_ab.py $name 4099 240 2>&1 | tail -1)
  b=$(PCX_LIB=$S/libpcx_$name.so python tools/experiments/spec_ab.py $name 4099 240 2>&1 | tail -1)
  [ "$a" == "$b" ] && echo "PARITY ok   $a" >> $OUT/ab.txt || { echo "PARITY FAIL $name"; echo "  stock: $a"; echo "  spec:  $b"; } >> $OUT/ab.txt
  echo "stock: $(python tools/generic_timing.py $fx 2>&1 | tail -1)" >> $OUT/ab.txt
  echo "spec:  $(PCX_LIB=$S/libpcx_$name.so python tools/generic_timing.py $fx 2>&1 | tail -1)" >> $OUT/ab.txt
done
cat $OUT/ab.txt
