#!/bin/bash
# Does the fractional last "round" of workgroups cost the headline kernel?  7 single-wave workgroups per CU
# (owner-code path) = 1792 slots; batch = slots x 64 x rounds.
run() { B=$1; shift; echo -n "batch $B $* : "; env "$@" python bench.py --batch $B --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); ms=d['roofline']['kernel_ms']; print(round(ms,4), 'ms', round(ms*1e6/$B,4), 'ns/env', round(d['roofline']['frac'],3))"; }
for B in 917504 1032192 1048576 1089536 1146880 1048576; do run $B PCX_X=0; done
for B in 1048576 917504 1179648; do run $B PCX_SM_CODES=0 PCX_WAVES_PER_CU=8; done
for B in 1048576 917504 1032192; do run $B PCX_SM_CODES=0 PCX_WAVES_PER_CU=7; done
