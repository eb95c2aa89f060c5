#!/bin/bash
# the run-time specialised build of pcx_generic_step: its tests, the whole GPU suite through the generic kernel, timing (JIT default / off)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_spec; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_generic_specialised.py -x -q -p no:cacheprovider > $OUT/tests_spec.txt 2>&1; echo "spec tests rc=$?"; tail -4 $OUT/tests_spec.txt
PCX_FORCE_GENERIC=1 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/suite_generic.txt 2>&1; echo "generic suite rc=$?"; tail -6 $OUT/suite_generic.txt
PCX_DEBUG=16 python tools/generic_timing.py warehouse_L0:262144 2>&1 | grep "pcx generic" | head -3
python tools/generic_timing.py > $OUT/timing_jit.txt 2>&1
PCX_GENERIC_JIT=0 python tools/generic_timing.py > $OUT/timing_table.txt 2>&1
echo "--- specialised (default)"; cat $OUT/timing_jit.txt; echo "--- table-driven (PCX_GENERIC_JIT=0)"; cat $OUT/timing_table.txt
ls pycolab_amd/csrc/jit_cache | head -3
