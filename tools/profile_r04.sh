#!/bin/bash
# Round-4 profiling recipe (run on the GPU box through gpurun).  The persistent launch shape's grid is the number of
# resident workgroups, not the batch, so launches can no longer be told apart by Grid_Size: every headline size gets
# runs of its own (`bench.py --batch B --no-other-configs`), each profiled four times -- kernel trace + stats, then
# three separate --pmc passes with --kernel-trace only (never mixed with other trace domains).
#   tools/profile_r04.sh  ->  gpurun_out/prof_r04/<batch>/{trace,pmc_write,pmc_fetch,pmc_sq}/ + summaries
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in 1048576 131072 262144; do
  D=$OUT/$B
  mkdir -p $D
  BENCH="python $ROOT/bench.py --batch $B --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o t -- $BENCH > $D/trace.log 2>&1
  grep '^{' $D/trace.log | tail -1 > $D/bench_under_rocprof.json
  for pass in "write WRITE_SIZE" "fetch FETCH_SIZE" "sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    set -- $pass
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D/pmc_$name -o p -- $BENCH > $D/pmc_$name.log 2>&1
  done
  find $D/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $D/kernel_stats.csv
  python $ROOT/tools/pmc_summary.py $D pcx_scrolly_maze_step > $D/pmc_summary.txt 2>&1
  echo "== batch $B"; grep pcx_scrolly $D/kernel_stats.csv | cut -c1-200 | head -4; cat $D/pmc_summary.txt
done
# the default line (headline + other configs) under the kernel trace: the rocprofv3 averages bench.py's HIP events must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default/trace -o t -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/default_trace.log 2>&1
grep '^{' $OUT/default_trace.log | tail -1 > $OUT/default_bench_under_rocprof.json
find $OUT/default/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/default_kernel_stats.csv
head -12 $OUT/default_kernel_stats.csv | cut -c1-200
