#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_call12
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_cropping.py tests/test_random_levels.py tests/test_reference_known_answers.py tests/test_postprocess.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED" $OUT/tests.log | tail -6
run() { echo -n "$* : "; env "$@" python bench.py --game scrolly_maze --batch ${B:-4096} --steps 2000 --warmup 100 --repeats 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms']*1000,2), 'us kernel')"; }
for e in 16 32 64; do run PCX_COOP_EPW=$e; done
run PCX_COOP_EPW=16 PCX_COOP_WAVES=4
for d in 1 2 4 7; do run PCX_DEBUG=$d; done
B=16384 run PCX_DEBUG=0
B=16384 run PCX_COOP_EPW=64
B=1024 run PCX_DEBUG=0
