#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { echo -n "$* : "; env "$@" python bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms (min %.4f max %.4f)' % (d['roofline']['kernel_ms'], min(d['repeats']['kernel_ms_all']), max(d['repeats']['kernel_ms_all'])))"; }
for w in 5 6 7 8 9 10 12; do run PCX_WAVES_PER_CU=$w; done
run PCX_SM_CODES=0
run PCX_WAVES_PER_CU=8
run PCX_WAVES_PER_WG=2 PCX_WGS_PER_CU=4
run PCX_WAVES_PER_WG=2 PCX_WGS_PER_CU=3
