#!/bin/bash
# Round 6: pcx_better_scrolly_step, the branch-free coin look-up against the round-5 loop ON ONE BOX: two libraries (PCX_LIB), runs
# alternating (the boxes of the pool differ by tens of per cent on this kernel, so nothing else compares).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_bs_ab
mkdir -p $OUT
cd $ROOT
{
for rep in 1 2 3; do
  echo "== round-5 loop (gpurun_variants/libpcx_round5_coin_loop.so), repetition $rep"
  PCX_LIB=$ROOT/gpurun_variants/libpcx_round5_coin_loop.so python tools/env_sweep.py --game better_scrolly_maze --batches 49152,65536,131072,262144 --steps 30 --variants "default" 2>&1 | grep -v amdgpu.ids
  echo "== branch-free look-up (csrc/libpcx.so), repetition $rep"
  python tools/env_sweep.py --game better_scrolly_maze --batches 49152,65536,131072,262144 --steps 30 --variants "default" 2>&1 | grep -v amdgpu.ids
done
echo "== the other boards, round-5 loop / branch-free"
for lib in $ROOT/gpurun_variants/libpcx_round5_coin_loop.so $ROOT/pycolab_amd/csrc/libpcx.so; do
  PCX_LIB=$lib python tools/env_sweep.py --fixture better_scrolly_maze_L1 --batches 262144 --steps 30 --variants "default" 2>&1 | grep -v amdgpu.ids
  PCX_LIB=$lib python tools/env_sweep.py --fixture better_scrolly_maze_L2 --batches 131072 --steps 30 --variants "default" 2>&1 | grep -v amdgpu.ids
  PCX_LIB=$lib python tools/env_sweep.py --fixture better_scrolly_custom_B --batches 262144 --steps 30 --variants "default" 2>&1 | grep -v amdgpu.ids
done
} > $OUT/r06_bs_coin_loop_same_box.txt 2>&1
cat $OUT/r06_bs_coin_loop_same_box.txt
