#!/bin/bash
# Round 3, second GPU call: parity of the rewritten logic phases (scrolly_maze probes / coin rows,
# table-driven kernel's mask probes), then same-box A/B against the library built from HEAD.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_call3
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log | head -2
bench() { python bench.py --no-cpu-baseline --no-other-configs --steps 100 --warmup 10 --repeats 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms (min %.4f max %.4f)' % (d['roofline']['kernel_ms'], min(d['repeats']['kernel_ms_all']), max(d['repeats']['kernel_ms_all'])))"; }
for rep in 1 2; do
  for lib in head new; do
    if [ $lib = head ]; then export PCX_LIB=$ROOT/gpurun_variants/libpcx_head.so; else unset PCX_LIB; fi
    echo -n "scrolly 1M    $lib: "; bench
    echo -n "scrolly 4096  $lib: "; bench --batch 4096 --steps 1000
    echo -n "scrolly 65536 $lib: "; bench --batch 65536 --steps 300
  done
done > $OUT/ab_scrolly.txt 2>&1
cat $OUT/ab_scrolly.txt
unset PCX_LIB
for v in head new genw4; do
  if [ $v = new ]; then unset PCX_LIB; else export PCX_LIB=$ROOT/gpurun_variants/libpcx_$v.so; fi
  echo "== generic kernel, library $v"
  timeout 300 python tools/generic_timing.py 2>&1 | grep -v amdgpu.ids
done > $OUT/generic_ab.txt 2>&1
cat $OUT/generic_ab.txt
unset PCX_LIB
bash tools/small_batch_ablation.sh > $OUT/small_ablation.txt 2>&1; cat $OUT/small_ablation.txt
