#!/bin/bash
# the three bench lines of tools/r06_final3.sh by themselves (after a change to bench.py that touches no kernel)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_final3
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/r06_bench_n1.json 2> $OUT/r06_bench_n1.err
python bench.py --steps 20 --warmup 5 > $OUT/r06_bench_n1_driver_flags.json 2> $OUT/r06_bench_n1_driver_flags.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o t -- python $ROOT/bench.py --no-cpu-baseline > $OUT/r06_bench_under_rocprof.json 2> $OUT/r06_bench_under_rocprof.err)
find $OUT/bench_trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/r06_bench_under_rocprof_kernel_stats.csv
rm -rf $OUT/bench_trace
tail -c 400 $OUT/r06_bench_n1.json
