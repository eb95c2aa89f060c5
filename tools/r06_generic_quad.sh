#!/bin/bash
# Round 6: pcx_generic_step with the state words quad-interleaved in HBM (16-byte write-back stores in the specialised build):
# parity (both builds, checkpoints, the forced-generic sweep), then same-box A/B against the row layout (PCX_GENERIC_QUAD=0).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_generic_quad
mkdir -p $OUT
cd $ROOT
Q="grep -v amdgpu.ids"
python -m pytest tests/test_generic_specialised.py tests/test_random_prefab_games.py tests/test_random_directive_games.py tests/test_generic_persistent.py tests/test_raise_parity.py tests/test_storytelling.py tests/test_checkpoint.py tests/test_postprocess.py tests/test_cropping.py -m gpu -q -x 2>&1 | tail -4 > $OUT/tests.txt
cat $OUT/tests.txt
PCX_FORCE_GENERIC=1 python -m pytest tests/test_hip_parity.py tests/test_random_levels.py tests/test_checkpoint.py -m gpu -q -x 2>&1 | tail -3 > $OUT/tests_forced.txt
cat $OUT/tests_forced.txt
{
export PCX_FORCE_GENERIC=1
V="quad;rows:!PCX_GENERIC_QUAD=0;quad_logic:PCX_DEBUG=2;rows_logic:!PCX_GENERIC_QUAD=0,PCX_DEBUG=2"
python tools/env_sweep.py --fixture warehouse_L0 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture directives_z_order --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture marauders_custom_A --batches 32768,262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture hello_world --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture warehouse_custom_B --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture walkers_room --batches 262144 --variants "$V" 2>&1 | $Q
echo "# phase timers (PCX_DEBUG=8), cycles per group of 64 environments"
for f in warehouse_L0 walkers_scroll_groups marauders_custom_A; do
  cf=0; [ $f = walkers_scroll_groups ] && cf=2
  for q in 1 0; do
    echo "== $f PCX_GENERIC_QUAD=$q"; PCX_GENERIC_QUAD=$q PCX_DEBUG=8 python tools/env_sweep.py --fixture $f --cardinal-fields $cf --batches 262144 --steps 40 --repeats 1 2>&1 | grep "pcx generic" | tail -1
  done
done
unset PCX_FORCE_GENERIC
} > $OUT/r06_generic_quad_sweep.txt 2>&1
cat $OUT/r06_generic_quad_sweep.txt
PCX_COOP_BELOW=0 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" > $OUT/gpu_suite_coop_below_0.txt; cat $OUT/gpu_suite_coop_below_0.txt
PCX_FORCE_GENERIC=1 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" > $OUT/gpu_suite_force_generic.txt; cat $OUT/gpu_suite_force_generic.txt
