#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_ego2
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_generic_specialised.py tests/test_random_prefab_games.py tests/test_random_directive_games.py tests/test_raise_parity.py tests/test_reference_known_answers.py tests/test_storytelling.py -m gpu -q 2>&1 | tail -3 > $OUT/tests.txt
PCX_FORCE_GENERIC=1 python -m pytest tests/test_hip_parity.py -m gpu -q -k "walkers or directives" 2>&1 | tail -3 >> $OUT/tests.txt
cat $OUT/tests.txt
{
export PCX_FORCE_GENERIC=1
V="auto"
for rep in 1 2; do
python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
done
python tools/env_sweep.py --fixture walkers_scroll_margins --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture walkers_hidden --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture warehouse_L0 --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture marauders_custom_A --batches 32768 --variants "$V" 2>&1 | grep -v amdgpu.ids
} > $OUT/r06_generic_ego_default.txt 2>&1
cat $OUT/r06_generic_ego_default.txt
