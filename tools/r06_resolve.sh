#!/bin/bash
# the sprite-resolve loop of the occlusion phase from registers in EVERY specialised build (auto) against from LDS where the sprites' state lives in LDS (-DPCX_X_RESOLVE_FROM_LDS), same box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_resolve
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_generic_specialised.py tests/test_random_prefab_games.py tests/test_random_directive_games.py -m gpu -q 2>&1 | tail -3 > $OUT/tests.txt
PCX_FORCE_GENERIC=1 PCX_GENERIC_JIT=1 timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -k "marauders or hello or better" 2>&1 | tail -3 >> $OUT/tests.txt
cat $OUT/tests.txt
{
export PCX_FORCE_GENERIC=1
V="auto;fromlds:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_RESOLVE_FROM_LDS"
for rep in 1 2; do
python tools/env_sweep.py --fixture marauders_custom_A --batches 32768 --variants "$V" 2>&1 | grep -v amdgpu.ids
done
python tools/env_sweep.py --fixture marauders --batches 32768 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture marauders_custom_A --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture hello_world --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture warehouse_L0 --batches 262144 --variants "$V" 2>&1 | grep -v amdgpu.ids
} > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
