#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_bs_scaling
mkdir -p $OUT
cd $ROOT
{
python tools/env_sweep.py --game better_scrolly_maze --batches 16384,32768,49152,65536,65600,73728,81920,98304,114688,131072,196608 --steps 30 --variants "default;w8:PCX_WAVES_PER_CU=8" 2>&1 | grep -v amdgpu.ids
} > $OUT/r06_bs_scaling.txt 2>&1
cat $OUT/r06_bs_scaling.txt
