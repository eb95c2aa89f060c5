#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_spec; mkdir -p $OUT
cd $ROOT
PCX_DEBUG=16 python tools/generic_timing.py 2>&1 | grep "pcx_generic\|waves per workgroup" | cut -c1-120 > $OUT/tuned.txt
cat $OUT/tuned.txt
echo "== table-driven build, tuned"
PCX_GENERIC_JIT=0 python tools/generic_timing.py 2>&1 | grep "pcx_generic" | cut -c1-120 | tee $OUT/tuned_table.txt
timeout 900 python -m pytest tests/test_generic_specialised.py tests/test_hip_parity.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3 | cut -c1-200
