#!/bin/bash
# Profiling recipe for the step kernel (run on the GPU box through gpurun).
# usage: tools/profile.sh <tag>     -> gpurun_out/prof_<tag>/{trace,pmc1..}/
# PMC passes are separate runs with --kernel-trace only (never mixed with
# other trace domains).
set -u
TAG=${1:-r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
i=0
for PMC in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
  "FETCH_SIZE" \
  "TCC_EA0_WRREQ_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_NORMAL_EVICT_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
done
find $OUT -name '*.csv' | head -40
