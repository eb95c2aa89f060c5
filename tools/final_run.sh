#!/bin/bash
# Round-end evidence run (through gpurun): GPU tests, smoke, default bench, the
# other BASELINE configs, rocprofv3 kernel trace + PMC passes.  Outputs under
# gpurun_out/final_<tag>/ ; copy what is to be judged into profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; cut -c1-400 $OUT/bench_n1.json
for cfg in "scrolly_maze 4096" "marauders 32768" "warehouse 262144" "hello_world 262144"; do
  set -- $cfg
  timeout 300 python bench.py --game $1 --batch $2 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/bench_configs.jsonl
done
python - <<PY
import json
for line in open('$OUT/bench_configs.jsonl'):
    d = json.loads(line)
    print(d['config']['workload'][:60], '| %.1f M env-steps/s | %.4f ms/step | roofline %.3f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac']))
PY
bash tools/profile.sh $TAG > $OUT/profile.log 2>&1
python tools/pmc_summary.py $ROOT/gpurun_out/prof_$TAG > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
find $ROOT/gpurun_out/prof_$TAG/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv; head -5 $OUT/kernel_stats.csv
