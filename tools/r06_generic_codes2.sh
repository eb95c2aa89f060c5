#!/bin/bash
# Round 6: pcx_generic_step with the render loop (owner codes / masks) measured by the engine's tuner next to the waves per workgroup:
# parity (both builds, the forced-generic sweep over every shipped game), then what the defaults settle on, same box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_generic_codes2
mkdir -p $OUT
cd $ROOT
Q="grep -v amdgpu.ids"
python -m pytest tests/test_generic_specialised.py tests/test_random_prefab_games.py tests/test_random_directive_games.py tests/test_generic_persistent.py tests/test_raise_parity.py tests/test_storytelling.py tests/test_postprocess.py tests/test_cropping.py tests/test_checkpoint.py -m gpu -q -x 2>&1 | tail -5 > $OUT/tests.txt
cat $OUT/tests.txt
PCX_FORCE_GENERIC=1 python -m pytest tests/test_hip_parity.py tests/test_random_levels.py -m gpu -q -x 2>&1 | tail -3 > $OUT/tests_forced.txt
cat $OUT/tests_forced.txt
PCX_FORCE_GENERIC=1 PCX_GENERIC_CODES=1 python -m pytest tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -3 > $OUT/tests_forced_codes.txt
cat $OUT/tests_forced_codes.txt
{
export PCX_FORCE_GENERIC=1
V="auto;codes:PCX_GENERIC_CODES=1;masks:PCX_GENERIC_CODES=0;old:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_NO_SPRITE_REGS,PCX_GENERIC_CODES=0"
python tools/env_sweep.py --fixture warehouse_L0 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture directives_z_order --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture marauders_custom_A --batches 32768,262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture hello_world --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture better_scrolly_maze_L1 --batches 65536 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture walkers_room --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture warehouse_custom_B --batches 262144 --variants "$V" 2>&1 | $Q
unset PCX_FORCE_GENERIC
} > $OUT/r06_generic_codes2_sweep.txt 2>&1
cat $OUT/r06_generic_codes2_sweep.txt
