#!/bin/bash
# Round 6, the evidence of the final tree from ONE box: bench.py's default line and the driver's flags, bench.py under rocprofv3
# (the same command: kernel stats), the per-row rocprofv3 passes (tools/profile_r06.sh), the GPU suite.
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
rm -rf $ROOT/gpurun_out/prof_r06
bash tools/profile_r06.sh > $OUT/r06_profile_log.txt 2>&1
python -m pytest tests -m gpu -q --durations=15 > $OUT/r06_gpu_suite.txt 2>&1
grep -E "passed|failed" $OUT/r06_gpu_suite.txt
tail -c 300 $OUT/r06_bench_n1.json
(timeout 300 python tools/ordeal_story_bench.py --batch 16384) 2>&1 | grep -v amdgpu.ids > $OUT/r06_ordeal_story.txt
cat $OUT/r06_ordeal_story.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 > $OUT/r06_smoke.txt
cat $OUT/r06_smoke.txt
