#!/bin/bash
# Round-4 end run: whole GPU suite, smoke, the default bench line (all rows).  ~10 GPU-minutes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_r04; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E " passed| failed" $OUT/pytest_gpu.log | tail -2
grep "^FAILED" $OUT/pytest_gpu.log | cut -c1-200 | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; cut -c1-400 $OUT/bench_n1.json; tail -2 $OUT/bench_n1.err | cut -c1-200
