#!/bin/bash
# Round 6: stream_windows with the window's feature-stack fields read once per window instead of once per iteration (a global load
# + s_waitcnt vmcnt(0) inside a store loop waits for every store before it): parity of the cropper tests, then same-box A/B of two
# libraries (PCX_LIB) on the example's three croppers fused into pcx_better_scrolly_step, and the fused croppers of the other kernels.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_windows
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_cropping.py tests/test_checkpoint.py tests/test_postprocess.py -m gpu -q 2>&1 | tail -3 > $OUT/tests.txt
cat $OUT/tests.txt
{
for rep in 1 2; do
  for lib in $ROOT/gpurun_variants/libpcx_before_window_hoist.so $ROOT/pycolab_amd/csrc/libpcx.so; do
    echo "== $lib"
    PCX_LIB=$lib python tools/fusion_bench.py win 2>&1 | grep -v amdgpu.ids
  done
done
for lib in $ROOT/gpurun_variants/libpcx_before_window_hoist.so $ROOT/pycolab_amd/csrc/libpcx.so; do
  echo "== $lib"
  PCX_LIB=$lib python tools/crop_features_bench.py 2>&1 | grep -v amdgpu.ids | tail -12
done
} > $OUT/r06_stream_windows_hoist.txt 2>&1
cat $OUT/r06_stream_windows_hoist.txt
