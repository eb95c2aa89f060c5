#!/bin/bash
# Round-3 evidence run (one GPU call): the whole GPU suite, the default bench line, the rocprofv3
# kernel trace + PMC passes of the same command, cropper / post-processor / fusion tables, the
# table-driven kernel's timings.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_final
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|FAILED" $OUT/suite.log | tail -4
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"
timeout 600 python bench.py --gpus 2 --oversubscribe --batch 524288 --gather --no-cpu-baseline > $OUT/bench_over2.json 2> $OUT/bench_over2.err; echo "bench oversubscribed rc=$?"
bash tools/profile_r02.sh r03 > $OUT/profile.log 2>&1; echo "profile rc=$?"; tail -30 $OUT/profile.log | cut -c1-200
timeout 900 python tools/post_bench.py > $OUT/post_kernels.md 2> $OUT/post.err; echo "post rc=$?"
timeout 600 python tools/fusion_bench.py > $OUT/fusion.txt 2>&1; cat $OUT/fusion.txt | grep -v amdgpu
timeout 600 python tools/generic_timing.py > $OUT/generic_timing.txt 2>&1; grep -v amdgpu $OUT/generic_timing.txt
PCX_FORCE_GENERIC=0 timeout 600 python tools/generic_timing.py warehouse_L0_unoccluded:262144 hello_world:262144 > $OUT/unocc_handwritten.txt 2>&1; grep -v amdgpu $OUT/unocc_handwritten.txt
bash tools/small_batch_ablation.sh > $OUT/small_ablation.txt 2>&1; cat $OUT/small_ablation.txt
bash tools/profile_generic_r03.sh > $OUT/profile_generic.log 2>&1; echo "generic profile rc=$?"
