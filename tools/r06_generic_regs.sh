#!/bin/bash
# Round 6: the specialised build of pcx_generic_step with the sprites' state in registers and the MazeWalker probes side by side
# (pcx_generic_kernel.h PCX_SREGS / PCX_PROBE_UNROLL): parity first, then same-box A/B against the round-5 build
# (PCX_GENERIC_SPEC_DEFS selects the variants: every one is its own cache entry).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_generic_regs
mkdir -p $OUT
cd $ROOT
Q="grep -v amdgpu.ids"
python -m pytest tests/test_generic_specialised.py tests/test_random_prefab_games.py tests/test_random_directive_games.py tests/test_generic_persistent.py -m gpu -q -x 2>&1 | tail -5 > $OUT/tests.txt
cat $OUT/tests.txt
{
export PCX_FORCE_GENERIC=1
OLD="-DPCX_X_NO_SPRITE_REGS -DPCX_X_NO_PROBE_UNROLL"
V="new;new_logic:PCX_DEBUG=2;wpe4:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_WPE=4;old:!PCX_GENERIC_SPEC_DEFS=$OLD;old_logic:!PCX_GENERIC_SPEC_DEFS=$OLD,PCX_DEBUG=2;regs_only:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_NO_PROBE_UNROLL;probes_only:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_NO_SPRITE_REGS"
python tools/env_sweep.py --fixture warehouse_L0 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture directives_z_order --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture marauders_custom_A --batches 32768,262144 --variants "$V" 2>&1 | $Q
echo "# phase timers (PCX_DEBUG=8), cycles per group of 64 environments"
for f in warehouse_L0 walkers_scroll_groups marauders_custom_A; do
  cf=0; [ $f = walkers_scroll_groups ] && cf=2
  for defs in "" "$OLD"; do
    echo "== $f defs='$defs'"; PCX_GENERIC_SPEC_DEFS="$defs" PCX_DEBUG=8 python tools/env_sweep.py --fixture $f --cardinal-fields $cf --batches 262144 --steps 40 --repeats 1 2>&1 | grep "pcx generic" | tail -1
  done
done
unset PCX_FORCE_GENERIC
echo "# scrolly_maze: the run-time build of a level of one's own next to the shipped levels' instances (same box)"
python tools/env_sweep.py --fixture scrolly_custom_H --batches 131072,1048576 --steps 40 --variants "jit" 2>&1 | $Q
python tools/env_sweep.py --fixture scrolly_maze_L1 --batches 131072,1048576 --steps 40 --variants "shipped_L1" 2>&1 | $Q
python tools/env_sweep.py --fixture scrolly_maze_L0 --batches 131072,1048576 --steps 40 --variants "shipped_L0" 2>&1 | $Q
} > $OUT/r06_generic_regs_sweep.txt 2>&1
tail -60 $OUT/r06_generic_regs_sweep.txt
