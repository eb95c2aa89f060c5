#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_ego
mkdir -p $OUT
cd $ROOT
{
export PCX_FORCE_GENERIC=1
V="auto;ego:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_EGO_UNROLL;auto_logic:PCX_DEBUG=2;ego_logic:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_EGO_UNROLL,PCX_DEBUG=2"
python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture walkers_scroll_margins --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture walkers_scroll_always --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
for defs in "" "-DPCX_X_EGO_UNROLL"; do
  echo "== walkers_scroll_groups defs='$defs'"; PCX_GENERIC_SPEC_DEFS="$defs" PCX_DEBUG=8 python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --steps 40 --repeats 1 2>&1 | grep "pcx generic" | tail -1
done
} > $OUT/r06_generic_ego_unroll.txt 2>&1
cat $OUT/r06_generic_ego_unroll.txt
