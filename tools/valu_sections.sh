#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS per launch for the PCX_DEBUG ablations of the step kernel.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/valu
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for d in 0 2 3 6 7; do
  PCX_DEBUG=$d rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/d$d -o p -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/d$d.log 2>&1
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open('$OUT/d$d/p_counter_collection.csv')):
    if 'pcx_' in row['Kernel_Name']: acc[row['Counter_Name']][row['Dispatch_Id']] += float(row['Counter_Value'])
print('DEBUG=$d', {k: round(sorted(v.values())[len(v)//2]/16384) for k,v in acc.items()})
PY
done
