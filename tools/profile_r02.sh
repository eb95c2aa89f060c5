#!/bin/bash
# Round-2 profiling recipe (run on the GPU box through gpurun).
#   tools/profile_r02.sh <tag>  ->  gpurun_out/prof_<tag>/{trace,pmc_write,pmc_fetch,pmc_sq}/ + summary files
# One command is profiled four times: `bench.py --steps 20 --warmup 3`, whose N=1
# line also measures BASELINE configs 2-4 (other_configs), so the kernel trace
# holds pcx_scrolly_maze_step (1,048,576 and 4,096 envs), pcx_marauders_step
# (32,768) and pcx_warehouse_step (262,144).  PMC passes are separate runs with
# --kernel-trace only (never mixed with other trace domains).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_under_rocprof.json
for pass in "write WRITE_SIZE" "fetch FETCH_SIZE" "sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  set -- $pass
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- $BENCH > $OUT/pmc_$name.log 2>&1
done
find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python $ROOT/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
head -8 $OUT/kernel_stats.csv | cut -c1-220
cat $OUT/pmc_summary.txt
