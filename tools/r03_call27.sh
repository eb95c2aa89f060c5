#!/bin/bash
OUT=gpurun_out/r03_call27; mkdir -p $OUT
run() { echo -n "$* : "; env "$@" python bench.py --game scrolly_maze --batch ${B:-4096} --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms']*1000,2), 'us kernel', round(d['ms_per_step']*1000,2), 'us/step')"; }
{
for r in 1 2; do run PCX_COOP_EPW=16; run PCX_COOP_EPW=8; done
B=2048 run PCX_COOP_EPW=16; B=2048 run PCX_COOP_EPW=8
B=8192 run PCX_COOP_EPW=16; B=8192 run PCX_COOP_EPW=8; B=8192 run PCX_COOP_EPW=32
run PCX_COOP_EPW=8 PCX_DEBUG=1; run PCX_COOP_EPW=8 PCX_DEBUG=2; run PCX_COOP_EPW=8 PCX_DEBUG=4; run PCX_COOP_EPW=8 PCX_DEBUG=7
} > $OUT/epw8.txt 2>&1
cat $OUT/epw8.txt
PCX_COOP_EPW=8 PCX_COOP_BELOW=1000000 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_cropping.py tests/test_postprocess.py -m gpu -q -x -k "scrolly" > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -4 $OUT/tests.log
