#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { echo -n "$* : "; env "$@" python bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms (min %.4f max %.4f)' % (d['roofline']['kernel_ms'], min(d['repeats']['kernel_ms_all']), max(d['repeats']['kernel_ms_all'])))"; }
for p in 1 2 3 1 2 3 5 9; do run PCX_SM_PASSES=$p; done
run PCX_SM_PASSES=2 PCX_WAVES_PER_CU=12
run PCX_SM_PASSES=3 PCX_WAVES_PER_CU=12
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "configuration_5 or full or beyond or 1048576 or million" 2>&1 | tail -2
