#!/bin/bash
# round 5, call 11: warehouse defaults (tests shrunk), the 8-rank oversubscribed bench line, then the profiling recipe
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call11; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_persistent_shapes.py -q -m gpu -x -k "warehouse" > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -3 $OUT/pytest.txt
timeout 600 python tools/wm_sweep.py --game warehouse --batches 131072,262144,524288,1048576 --variants pw0,auto,w8k4,w1k0x4,w8k4x2,w2k1x4 > $OUT/wm_sweep.txt 2>&1; grep -v amdgpu.ids $OUT/wm_sweep.txt | tail -26
timeout 600 python bench.py --gpus 8 --oversubscribe --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_oversubscribed_8ranks.json 2> $OUT/bench_oversubscribed_8ranks.err; tail -c 1200 $OUT/bench_oversubscribed_8ranks.json
bash tools/profile_r05.sh > $OUT/profile.txt 2>&1; tail -60 $OUT/profile.txt
