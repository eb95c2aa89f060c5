#!/bin/bash
# Round 6: pcx_generic_step's mask-composing render loop with the things' descriptors as constants in the specialised build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_generic_masks
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_generic_specialised.py tests/test_random_prefab_games.py tests/test_random_directive_games.py tests/test_postprocess.py tests/test_cropping.py -m gpu -q -x 2>&1 | tail -3 > $OUT/tests.txt
PCX_FORCE_GENERIC=1 PCX_GENERIC_CODES=0 python -m pytest tests/test_hip_parity.py tests/test_random_levels.py -m gpu -q -x 2>&1 | tail -3 >> $OUT/tests.txt
cat $OUT/tests.txt
{
export PCX_FORCE_GENERIC=1
for lib in $ROOT/gpurun_variants/libpcx_round5_coin_loop.so $ROOT/pycolab_amd/csrc/libpcx.so; do
  echo "== $lib"
  PCX_LIB=$lib python tools/env_sweep.py --fixture marauders_custom_A --batches 32768,262144 --variants "auto;masks:PCX_GENERIC_CODES=0" 2>&1 | grep -v amdgpu.ids
  PCX_LIB=$lib python tools/env_sweep.py --fixture better_scrolly_maze_L1 --batches 65536 --variants "auto" 2>&1 | grep -v amdgpu.ids
  PCX_LIB=$lib python tools/env_sweep.py --fixture warehouse_L0 --batches 262144 --variants "auto;masks:PCX_GENERIC_CODES=0" 2>&1 | grep -v amdgpu.ids
  PCX_LIB=$lib python tools/env_sweep.py --fixture warehouse_L0_unoccluded --batches 262144 --variants "auto" 2>&1 | grep -v amdgpu.ids
  PCX_LIB=$lib python tools/env_sweep.py --fixture hello_world --batches 262144 --variants "auto;masks:PCX_GENERIC_CODES=0" 2>&1 | grep -v amdgpu.ids
  PCX_LIB=$lib python tools/env_sweep.py --fixture marauders --batches 32768 --variants "auto" 2>&1 | grep -v amdgpu.ids
done
unset PCX_FORCE_GENERIC
} > $OUT/r06_generic_masks_sweep.txt 2>&1
cat $OUT/r06_generic_masks_sweep.txt
