#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { echo -n "B=$B $* : "; env "$@" python bench.py --game scrolly_maze --batch $B --steps 1000 --warmup 100 --repeats 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms']*1000,2), 'us kernel')"; }
for B in 8192 16384 32768 65536; do for e in 16 32 64; do run PCX_COOP_EPW=$e; done; done
B=65536 run PCX_COOP_BELOW=0
B=32768 run PCX_COOP_BELOW=0
