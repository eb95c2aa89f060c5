#!/bin/bash
# scratch driver for one gpurun call (overwritten per experiment)
OUT=gpurun_out/r03_call38; mkdir -p $OUT
timeout 900 python -m pytest tests/test_cropping.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -12 $OUT/tests.log
python bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step'])
for c in d['other_configs']: print(c['workload'], c['ms_per_step'])"
