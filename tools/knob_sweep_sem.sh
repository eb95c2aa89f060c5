#!/bin/bash
# PCX_STREAM_SLOTS experiment (headline kernel, 1,048,576 envs): resident waves per CU x streaming slots per CU.
run() { echo -n "$* : "; env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],4))"; }
run PCX_SM_CODES=1
run PCX_SM_CODES=0 PCX_WAVES_PER_CU=8
run PCX_SM_CODES=0 PCX_WAVES_PER_CU=8 PCX_STREAM_SLOTS=8
run PCX_SM_CODES=0 PCX_WAVES_PER_CU=10
run PCX_SM_CODES=0 PCX_WAVES_PER_CU=10 PCX_STREAM_SLOTS=8
run PCX_SM_CODES=0 PCX_WAVES_PER_CU=10 PCX_STREAM_SLOTS=7
run PCX_SM_CODES=0 PCX_WAVES_PER_CU=12
run PCX_SM_CODES=0 PCX_WAVES_PER_CU=12 PCX_STREAM_SLOTS=8
run PCX_SM_CODES=0 PCX_WAVES_PER_CU=12 PCX_STREAM_SLOTS=7
run PCX_SM_CODES=0 PCX_WAVES_PER_CU=12 PCX_STREAM_SLOTS=6
run PCX_SM_CODES=0 PCX_WAVES_PER_CU=15 PCX_STREAM_SLOTS=8
run PCX_SM_CODES=0 PCX_WAVES_PER_CU=15 PCX_STREAM_SLOTS=10
run PCX_SM_CODES=1
