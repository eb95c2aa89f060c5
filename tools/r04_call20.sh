#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_spec; mkdir -p $OUT
cd $ROOT
PCX_DEBUG=16 PCX_FORCE_GENERIC=1 python tools/generic_timing.py marauders_unoccluded:32768 marauders:4096 2>&1 | grep -v amdgpu.ids | tail -6
timeout 1200 python -m pytest tests/test_generic_specialised.py -q -p no:cacheprovider > $OUT/tests_spec.txt 2>&1; echo "spec tests rc=$?"; tail -8 $OUT/tests_spec.txt
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x --deselect tests/test_generic_specialised.py > $OUT/suite.txt 2>&1; echo "suite rc=$?"; tail -6 $OUT/suite.txt
