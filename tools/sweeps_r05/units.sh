#!/bin/bash
# round 5, call 17: work units of 32 / 16 environments with the compiled-in instance (finer granularity at the end of a launch)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call17; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_persistent_shapes.py -q -m gpu -x -k "shapes_match or small_units" > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt; tail -3 $OUT/pytest.txt
timeout 900 python tools/ps_sweep.py --batches 131072,262144,1048576 --steps 100 --repeats 3 --prof \
  --variants unbaked \
  --extra "def:PCX_SM_TUNE=0,u32:PCX_SM_UNIT=32,u32w6k3:PCX_SM_UNIT=32+PCX_SM_WAVES=6+PCX_SM_LOCK=3,u32w8k4:PCX_SM_UNIT=32+PCX_SM_WAVES=8+PCX_SM_LOCK=4,u32w4k2x2:PCX_SM_UNIT=32+PCX_SM_WAVES=4+PCX_SM_LOCK=2+PCX_SM_PER_CU=2,u32w6k2:PCX_SM_UNIT=32+PCX_SM_WAVES=6+PCX_SM_LOCK=2,u16w8k4:PCX_SM_UNIT=16+PCX_SM_WAVES=8+PCX_SM_LOCK=4,t1u32:PCX_SM_TAIL=1+PCX_SM_TAIL_UNIT=32,t2u32:PCX_SM_TAIL=2+PCX_SM_TAIL_UNIT=32,t1u16:PCX_SM_TAIL=1+PCX_SM_TAIL_UNIT=16" \
  --out $OUT/ps_sweep.json > $OUT/ps_sweep.txt 2>&1
grep -v "amdgpu.ids" $OUT/ps_sweep.txt | grep -v "^ *\[" | tail -36
grep "^ *\[\(def\|u32\|u32w6k3\|t1u32\)\]" $OUT/ps_sweep.txt | cut -c1-220 | tail -8
