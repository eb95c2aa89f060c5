#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: per-kernel mean of each counter."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else 'pcx_'
acc = collections.defaultdict(list)
for path in sorted(glob.glob(root + '/pmc*/p_counter_collection.csv')):
  per_dispatch = collections.defaultdict(float)
  for row in csv.DictReader(open(path)):
    if match not in row['Kernel_Name']:
      continue
    per_dispatch[(row['Dispatch_Id'], row['Counter_Name'])] += float(row['Counter_Value'])
  for (_, name), v in per_dispatch.items():
    acc[name].append(v)
for name, vals in acc.items():
  vals = vals[len(vals) // 4:]  # drop warmup/reset dispatches
  print('%-28s n=%3d mean=%16.1f' % (name, len(vals), sum(vals) / len(vals)))
