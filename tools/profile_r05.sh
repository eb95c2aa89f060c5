#!/bin/bash
# Round-5 profiling recipe (run on the GPU box through gpurun): as tools/profile_r04.sh -- every size gets runs of its own
# (`bench.py --batch B --no-other-configs`), each profiled four times: kernel trace + stats, then three separate --pmc passes
# with --kernel-trace only (never mixed with other trace domains).  PCX_SM_TUNE=0: the launch-shape tuner would put four
# different grids into one average; the default shape is what is profiled.
#   tools/profile_r05.sh  ->  gpurun_out/prof_r05/<game>_<batch>/{trace,pmc_write,pmc_fetch,pmc_sq}/ + summaries
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PCX_SM_TUNE=0
for spec in scrolly_maze:1048576:pcx_scrolly_maze_step scrolly_maze:131072:pcx_scrolly_maze_step warehouse:262144:pcx_warehouse_step; do
  game=${spec%%:*}; rest=${spec#*:}; B=${rest%%:*}; kern=${rest#*:}
  D=$OUT/${game}_$B
  mkdir -p $D
  BENCH="python $ROOT/bench.py --game $game --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o t -- $BENCH > $D/trace.log 2>&1
  grep '^{' $D/trace.log | tail -1 > $D/bench_under_rocprof.json
  for pass in "write WRITE_SIZE" "fetch FETCH_SIZE" "sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    set -- $pass
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D/pmc_$name -o p -- $BENCH > $D/pmc_$name.log 2>&1
  done
  find $D/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $D/kernel_stats.csv
  python $ROOT/tools/pmc_summary.py $D $kern > $D/pmc_summary.txt 2>&1
  echo "== $game $B"; grep pcx_ $D/kernel_stats.csv | cut -c1-220 | head -4; cat $D/pmc_summary.txt
  rm -rf $D/trace $D/pmc_write $D/pmc_fetch $D/pmc_sq   # (the raw traces are large; the summaries are what is kept)
done
unset PCX_SM_TUNE
# the default line (headline + other configs) under the kernel trace: the rocprofv3 averages bench.py's HIP events must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default/trace -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/default_trace.log 2>&1
grep '^{' $OUT/default_trace.log | tail -1 > $OUT/default_bench_under_rocprof.json
find $OUT/default/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/default_kernel_stats.csv
rm -rf $OUT/default/trace
head -14 $OUT/default_kernel_stats.csv | cut -c1-200
