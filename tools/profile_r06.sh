#!/bin/bash
# Round-6 profiling recipe (run on the GPU box through gpurun).  EVERY row of bench.py's line gets runs of its own
# (tools/row_bench.py: the only step kernel in a trace is the row's), each profiled three times: kernel trace + stats, then two
# separate --pmc passes (WRITE_SIZE, FETCH_SIZE) with --kernel-trace only -- never mixed with other trace domains, as
# MI355X_MICROARCH.md prescribes.  The launch-shape tuners stay ON: a row's record names the launch shape its own profiled run
# settled on (the row's JSON), and bench.py uses a record only for a run that took the same kernel in the same shape
# (VERDICT r5 weak #6).  The headline row also gets an SQ pass.
#   tools/profile_r06.sh [row ...]  ->  gpurun_out/prof_r06/<row>/{row.json,kernel_stats.csv,pmc_summary.txt}
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ROWS=${*:-headline scrolly_131072 scrolly_262144 scrolly_custom_H_131072 scrolly_L1_131072 scrolly_4096 marauders_32768 marauders_262144 warehouse_262144 better_scrolly_65536 hello_world_1048576 marauders_custom_A walkers warehouse_generic ordeal_kansas}
for row in $ROWS; do
  D=$OUT/$row
  mkdir -p $D
  RUN="python $ROOT/tools/row_bench.py $row"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o t -- $RUN > $D/trace.log 2>&1
  grep '^{' $D/trace.log | tail -1 > $D/row.json
  passes=("write WRITE_SIZE" "fetch FETCH_SIZE")
  [ $row = headline ] && passes+=("sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES")
  for pass in "${passes[@]}"; do
    set -- $pass
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D/pmc_$name -o p -- $RUN > $D/pmc_$name.log 2>&1
    grep '^{' $D/pmc_$name.log | tail -1 > $D/row_pmc_$name.json
  done
  find $D/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $D/kernel_stats.csv
  python $ROOT/tools/pmc_summary.py $D pcx_ > $D/pmc_summary.txt 2>&1
  echo "== $row"; python -c "import json;r=json.load(open('$D/row.json'));print(r['kernel'],r['launch_shape'],'%.4f ms'%r['ms_per_step'],'%.3f'%r['hbm_frac'])"
  grep "pcx_.*_step" $D/kernel_stats.csv | cut -d, -f1-4 | head -3; grep "_step" $D/pmc_summary.txt
  rm -rf $D/trace $D/pmc_write $D/pmc_fetch $D/pmc_sq   # (the raw traces are large; the summaries are what is kept)
done
