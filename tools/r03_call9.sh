#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_call9
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in head new; do
  if [ $lib = head ]; then export PCX_LIB=$ROOT/gpurun_variants/libpcx_head.so; else unset PCX_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$lib -o t -- python $ROOT/tools/crop_profile.py > $OUT/trace_$lib.log 2>&1
  echo "== $lib"; python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for path in glob.glob('$OUT/trace_$lib/**/t_kernel_trace.csv', recursive=True):
  for row in csv.DictReader(open(path)):
    if 'pcx_crop' in row['Kernel_Name']:
      acc[(row['Kernel_Name'].split('(')[0].split('::')[-1], row['Grid_Size'])].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
for k, v in sorted(acc.items()):
  v = v[len(v)//4:]
  print('%-18s grid %-10s n=%3d mean %.1f us' % (k[0], k[1], len(v), sum(v) / len(v) / 1e3))
PY
done
unset PCX_LIB
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python $ROOT/tools/crop_profile.py > $OUT/pmc_$c.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT pcx_crop
