#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_call14
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|FAILED" $OUT/suite.log | tail -6
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('$OUT/bench_n1.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_achievable'))
for o in d['other_configs']: print(o['workload'], round(o['ms_per_step']*1000,2),'us', round(o['hbm_frac'],3))
PY
