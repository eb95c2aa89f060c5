#!/bin/bash
# Round 6: pcx_better_scrolly_step with the coin look-up of its streaming loop branch-free (four mask words requested at once, the
# dword's ids fetched an iteration ahead): parity, then timing (the same box as the round-5 loop is whatever r06_bs_waves ran on --
# the boxes differ, so the workgroups-per-CU sweep is repeated here with the new loop).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_bs_coins
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_hip_parity.py tests/test_gate_digests.py tests/test_random_levels.py tests/test_cropping.py tests/test_postprocess.py tests/test_checkpoint.py tests/test_reference_live.py -m gpu -q -k "better" 2>&1 | tail -3 > $OUT/tests.txt
PCX_COOP_BELOW=0 python -m pytest tests/test_hip_parity.py tests/test_random_levels.py tests/test_cropping.py -m gpu -q -k "better" 2>&1 | tail -3 >> $OUT/tests.txt
cat $OUT/tests.txt
{
V="default;w2:PCX_WAVES_PER_CU=2;w3:PCX_WAVES_PER_CU=3;w4:PCX_WAVES_PER_CU=4;w5:PCX_WAVES_PER_CU=5;w6:PCX_WAVES_PER_CU=6;w8:PCX_WAVES_PER_CU=8;w12:PCX_WAVES_PER_CU=12;logic:PCX_DEBUG=2"
python tools/env_sweep.py --game better_scrolly_maze --batches 65536,131072,262144 --steps 30 --variants "$V" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture better_scrolly_maze_L1 --batches 262144 --steps 30 --variants "default;w4:PCX_WAVES_PER_CU=4;w8:PCX_WAVES_PER_CU=8;w12:PCX_WAVES_PER_CU=12" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture better_scrolly_maze_L2 --batches 131072 --steps 30 --variants "default;w4:PCX_WAVES_PER_CU=4;w8:PCX_WAVES_PER_CU=8;w12:PCX_WAVES_PER_CU=12" 2>&1 | grep -v amdgpu.ids
python tools/env_sweep.py --fixture better_scrolly_custom_B --batches 262144 --steps 30 --variants "default" 2>&1 | grep -v amdgpu.ids
} > $OUT/r06_bs_coins_sweep.txt 2>&1
cat $OUT/r06_bs_coins_sweep.txt
