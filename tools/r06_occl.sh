#!/bin/bash
# the occlusion phase of pcx_generic_step with its LDS reads batched (default) against word by word (-DPCX_X_OCCL_WORD_BY_WORD), specialised build, same box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_occl
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_generic_specialised.py tests/test_generic_persistent.py tests/test_random_prefab_games.py tests/test_random_directive_games.py tests/test_ordeal.py tests/test_reference_known_answers.py -m gpu -q 2>&1 | tail -3 > $OUT/tests.txt
PCX_FORCE_GENERIC=1 timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_cropping.py tests/test_postprocess.py tests/test_checkpoint.py -m gpu -q 2>&1 | tail -3 >> $OUT/tests.txt
cat $OUT/tests.txt
{
export PCX_FORCE_GENERIC=1
V="auto;wordbyword:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_OCCL_WORD_BY_WORD"
python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture warehouse_L0 --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture marauders_custom_A --batches 32768 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture hello_world --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture better_scrolly_maze_L1 --batches 65536 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture directives_z_order --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture walkers_scroll_margins --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
} > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
