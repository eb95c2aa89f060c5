#!/bin/bash
# Round 6, last validations on one box: a store-priority experiment on pcx_generic_step, the GPU suite with every kernel forced into
# its large-batch shape and with every shipped game but scrolly_maze through pcx_generic_step, the N > 1 path oversubscribed.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_validate
mkdir -p $OUT
cd $ROOT
Q="grep -v amdgpu.ids"
{
export PCX_FORCE_GENERIC=1
V="auto;prio:!PCX_GENERIC_SPEC_DEFS=-DPCX_X_WB_PRIO"
python tools/env_sweep.py --fixture warehouse_L0 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture walkers_scroll_groups --cardinal-fields 2 --batches 262144 --variants "$V" 2>&1 | $Q
python tools/env_sweep.py --fixture marauders_custom_A --batches 32768 --variants "$V" 2>&1 | $Q
unset PCX_FORCE_GENERIC
} > $OUT/r06_generic_store_priority.txt 2>&1
cat $OUT/r06_generic_store_priority.txt
PCX_COOP_BELOW=0 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/gpu_suite_coop_below_0.txt; cat $OUT/gpu_suite_coop_below_0.txt
PCX_FORCE_GENERIC=1 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/gpu_suite_force_generic.txt; cat $OUT/gpu_suite_force_generic.txt
python bench.py --gpus 2 --oversubscribe --no-cpu-baseline > $OUT/r06_bench_oversubscribed_2ranks.json 2> $OUT/o2.err; tail -c 400 $OUT/r06_bench_oversubscribed_2ranks.json
python bench.py --gpus 8 --oversubscribe --no-cpu-baseline > $OUT/r06_bench_oversubscribed_8ranks.json 2> $OUT/o8.err; tail -c 400 $OUT/r06_bench_oversubscribed_8ranks.json
