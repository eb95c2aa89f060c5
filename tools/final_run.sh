#!/bin/bash
# Round-end evidence run (through gpurun): GPU tests, smoke, rocprofv3 kernel trace + PMC passes of the default
# bench command, the default bench line, the cropper / post-processor table, unshipped-level timing.
# Outputs under gpurun_out/final_<tag>/ and gpurun_out/prof_<tag>/ ; copy what is to be judged into profiles/
# (kernel_stats.csv, bench_under_rocprof.json, pmc_summary.txt -> tools/traffic_records.py, bench_n1.json, post.md).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/profile_r02.sh $TAG > $OUT/profile.log 2>&1
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; cut -c1-300 $OUT/bench_n1.json
python tools/post_bench.py --steps 200 > $OUT/post.md 2> $OUT/post.err; wc -l $OUT/post.md
bash tools/custom_level_timing.sh 2>&1 | grep -v amdgpu.ids > $OUT/custom_levels.txt
bash tools/window_phase.sh 2>&1 | grep -v amdgpu.ids > $OUT/window_phase.txt
