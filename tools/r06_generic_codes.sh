#!/bin/bash
# Round 6: pcx_generic_step rendering from owner codes (plain steps), the sprites' state in registers (specialised build, up to
# eight sprites), the compact LDS layout: parity first, then same-box A/B against the mask-composing loop and the round-5 build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_generic_codes
mkdir -p $OUT
cd $ROOT
Q="grep -v amdgpu.ids"
python -m pytest tests/test_generic_specialised.py tests/test_random_prefab_games.py tests/test_random_directive_games.py tests/test_generic_persistent.py tests/test_raise_parity.py tests/test_storytelling.py tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -5 > $OUT/tests.txt
cat $OUT/tests.txt
PCX_FORCE_GENERIC=1 python -m pytest tests/test_hip_parity.py tests/test_random_levels.py -m gpu -q -x 2>&1 | tail -3 > $OUT/tests_forced.txt
cat $OUT/tests_forced.txt
{
export PCX_FORCE_GENERIC=1
OLD="-DPCX_X_NO_SPRITE_REGS"
V="new;new_logic:PCX_DEBUG=2;masks:PCX_GENERIC_CODES=0;wpe4:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_WPE=4;old:!PCX_GENERIC_SPEC_DEFS=$OLD,PCX_GENERIC_CODES=0;old_codes:!PCX_GENERIC_SPEC_DEFS=$OLD;wb4:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_WB4;w1:PCX_GENERIC_WAVES=1;w2:PCX_GENERIC_WAVES=2;w4:PCX_GENERIC_WAVES=4"
python tools/env_sweep.py --fixture warehouse_L0 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture directives_z_order --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture marauders_custom_A --batches 32768,262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture hello_world --batches 262144 --variants "new;masks:PCX_GENERIC_CODES=0" 2>&1 | $Q
python tools/env_sweep.py --fixture better_scrolly_maze_L1 --batches 65536 --variants "new;masks:PCX_GENERIC_CODES=0" 2>&1 | $Q
echo "# phase timers (PCX_DEBUG=8), cycles per group of 64 environments"
for f in warehouse_L0 walkers_scroll_groups marauders_custom_A; do
  cf=0; [ $f = walkers_scroll_groups ] && cf=2
  for defs in "" "-DPCX_X_WB4"; do
    echo "== $f defs='$defs'"; PCX_GENERIC_SPEC_DEFS="$defs" PCX_DEBUG=8 python tools/env_sweep.py --fixture $f --cardinal-fields $cf --batches 262144 --steps 40 --repeats 1 2>&1 | grep "pcx generic" | tail -1
  done
done
unset PCX_FORCE_GENERIC
} > $OUT/r06_generic_codes_sweep.txt 2>&1
tail -80 $OUT/r06_generic_codes_sweep.txt
