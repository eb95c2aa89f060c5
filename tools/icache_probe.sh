#!/bin/bash
# Instruction-cache behaviour of the step kernels at a small batch (latency-bound logic phase).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|inst_cache|SQC_" | head -40 > $OUT/avail.txt
cat $OUT/avail.txt | cut -c1-160 | head -30
for B in 4096 1048576; do
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/b$B -o p -- python $ROOT/bench.py --batch $B --steps 20 --warmup 3 --no-cpu-baseline > $OUT/b$B.log 2>&1
  echo "== batch $B"; python $ROOT/tools/pmc_summary.py $OUT/b$B 2>/dev/null || python - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for path in glob.glob('$OUT/b$B/**/p_counter_collection.csv', recursive=True):
  per = collections.defaultdict(float)
  for row in csv.DictReader(open(path)):
    if 'pcx_' not in row['Kernel_Name']: continue
    per[(row['Dispatch_Id'], row['Counter_Name'])] += float(row['Counter_Value'])
  for (_, n), v in per.items(): acc[n].append(v)
for n, v in acc.items():
  v = v[len(v)//4:]; print('%-24s n=%3d mean=%14.1f' % (n, len(v), sum(v)/len(v)))
PY
done
