#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_reference_known_answers.py -m gpu -q -x 2>&1 | tail -2
run() { echo -n "$* : "; env "$@" python bench.py --game scrolly_maze --batch ${B:-4096} --steps 2000 --warmup 100 --repeats 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms']*1000,2), 'us kernel')"; }
for d in 0 1 2 4 7; do run PCX_DEBUG=$d; done
B=16384 run PCX_DEBUG=0
B=1024 run PCX_DEBUG=0
B=256 run PCX_DEBUG=0
