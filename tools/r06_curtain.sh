#!/bin/bash
# update_curtain with four rows' pattern words in flight (default) against row by row (-DPCX_X_CURTAIN_ROW_BY_ROW), specialised build, same box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_curtain
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_generic_specialised.py tests/test_random_prefab_games.py tests/test_raise_parity.py tests/test_reference_known_answers.py -m gpu -q 2>&1 | tail -3 > $OUT/tests.txt
PCX_FORCE_GENERIC=1 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_cropping.py -m gpu -q -k "walkers or scroll" 2>&1 | tail -3 >> $OUT/tests.txt
cat $OUT/tests.txt
{
export PCX_FORCE_GENERIC=1
V="auto;rowbyrow:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_CURTAIN_ROW_BY_ROW"
for rep in 1 2; do
python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
done
python tools/env_sweep.py --fixture walkers_scroll_margins --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture walkers_scroll_always --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture walkers_hidden --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
} > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
