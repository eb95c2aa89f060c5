#!/bin/bash
# Round 6: pcx_better_scrolly_step, single-wave workgroups per CU (the LDS pad decides how many share a CU).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_bs_waves
mkdir -p $OUT
cd $ROOT
{
V="default;w2:PCX_WAVES_PER_CU=2;w3:PCX_WAVES_PER_CU=3;w4:PCX_WAVES_PER_CU=4;w5:PCX_WAVES_PER_CU=5;w6:PCX_WAVES_PER_CU=6;w8:PCX_WAVES_PER_CU=8;w12:PCX_WAVES_PER_CU=12;logic:PCX_DEBUG=2"
python tools/env_sweep.py --game better_scrolly_maze --batches 65536,131072,262144 --steps 30 --variants "$V" 2>&1 | grep -v amdgpu.ids
} > $OUT/r06_bs_waves_sweep.txt 2>&1
cat $OUT/r06_bs_waves_sweep.txt
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" > $OUT/gpu_suite.txt; cat $OUT/gpu_suite.txt
