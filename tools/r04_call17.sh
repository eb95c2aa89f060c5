#!/bin/bash
# dump the specialisation headers of the generic_timing fixtures (GenericBackend::spec_header)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_spec; mkdir -p $OUT
cd $ROOT
for fx in warehouse_L0 marauders_custom_A marauders walkers_scroll_groups directives_z_order walkers_room hello_world warehouse_L0_unoccluded better_scrolly_custom_B marauders_unoccluded; do
  PCX_GENERIC_DUMP_SPEC=$OUT/$fx.h python tools/generic_timing.py $fx:4096 2>&1 | tail -1
done
ls -la $OUT
