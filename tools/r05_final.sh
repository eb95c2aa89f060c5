#!/bin/bash
# round 5, evidence run: the whole GPU suite, smoke, then bench.py as the driver runs it (and once with its own defaults)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_final; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -25 $OUT/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err; tail -c 600 $OUT/bench_driver_flags.json
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -c 300 $OUT/bench_n1.json
