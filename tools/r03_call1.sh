#!/bin/bash
# Round 3, first GPU call: the new tests, the whole GPU suite, the hardened bench line (N=1 and the
# oversubscribed N=2 path), and where the table-driven kernel stands before this round's work.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_call1
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_distributed.py tests/test_storytelling.py tests/test_postprocess.py -m gpu -x -q > $OUT/new_tests.log 2>&1; echo "new tests rc=$?"; tail -3 $OUT/new_tests.log
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"; cut -c1-1500 $OUT/bench_n1.json
timeout 600 python bench.py --gpus 2 --oversubscribe --batch 524288 --gather --no-cpu-baseline > $OUT/bench_over2.json 2> $OUT/bench_over2.err; echo "bench over rc=$?"; cut -c1-1800 $OUT/bench_over2.json; tail -3 $OUT/bench_over2.err
timeout 600 python tools/generic_timing.py > $OUT/generic_timing.txt 2>&1; cat $OUT/generic_timing.txt
