#!/bin/bash
# rocprofv3 kernel statistics of the table-driven kernel at the other BASELINE configs.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_generic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "marauders 32768" "warehouse 262144" "hello_world 262144"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -o t -- python $ROOT/bench.py --game $1 --batch $2 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/$1.log 2>&1
  echo "== $1 batch $2: bench line, then rocprofv3 kernel stats"
  grep '^{' $OUT/$1.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py: %.4f ms per step (HIP events %.4f ms)' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  find $OUT/$1 -name '*kernel_stats.csv' | head -1 | xargs head -3 | cut -c1-200
done
