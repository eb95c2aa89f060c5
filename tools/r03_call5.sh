#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_call6
mkdir -p $OUT
cd $ROOT
echo skip suite
cd /tmp && export TMPDIR=/tmp
for pass in "a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "b SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY" "c SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU"; do
  set -- $pass; name=$1; shift
  PCX_FORCE_GENERIC=1 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- python $ROOT/tools/generic_timing.py warehouse_L0:262144 walkers_scroll_groups:262144 marauders:32768 > $OUT/pmc_$name.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT pcx_generic > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
