#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs under <root>/pmc*/: per kernel
and per launch size (grid), the mean of each counter over the step launches
(the first quarter of every group -- warm-up and reset launches -- dropped)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else 'pcx_'
acc = collections.defaultdict(list)
for path in sorted(glob.glob(root + '/pmc*/**/*counter_collection.csv', recursive=True)):
  per_dispatch = collections.OrderedDict()
  for row in csv.DictReader(open(path)):
    if match not in row['Kernel_Name']:
      continue
    short = row['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1]
    key = (short, row.get('Grid_Size', '?'), row['Dispatch_Id'], row['Counter_Name'])
    per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row['Counter_Value'])
  for (short, grid, _, name), v in per_dispatch.items():
    acc[(short, grid, name)].append(v)
for (short, grid, name), vals in sorted(acc.items()):
  vals = vals[len(vals) // 4:]
  print('%-24s grid %-10s %-22s n=%4d mean=%18.1f' % (short, grid, name, len(vals), sum(vals) / len(vals)))
