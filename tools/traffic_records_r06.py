#!/usr/bin/env python3
"""profiles/hbm_traffic.json from tools/profile_r06.sh's per-row output: HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE
(both in KiB; gfx950 reports half of a coalesced read stream -- MI355X_MICROARCH.md, HBM section), per row of bench.py's
line, each record naming the kernel AND the launch shape its profiled run took (bench.py matches on both).
  python tools/traffic_records_r06.py gpurun_out/prof_r06 profiles/r06_pmc_summary.txt > profiles/hbm_traffic.json"""
import json
import os
import re
import sys

root, source = sys.argv[1], sys.argv[2]
records = []
for row in sorted(os.listdir(root)):
  d = os.path.join(root, row)
  try:
    r = json.load(open(os.path.join(d, 'row.json')))
    shapes = {json.load(open(os.path.join(d, 'row_pmc_%s.json' % p)))['launch_shape'] for p in ('write', 'fetch')} | {r['launch_shape']}
  except Exception:  # pylint: disable=broad-except
    continue
  vals = {}
  for line in open(os.path.join(d, 'pmc_summary.txt')):
    m = re.match(r'(\S+)\s+grid (\S+)\s+(\S+)\s+n=\s*(\d+) mean=\s*([\d.]+)', line)
    if m and m.group(1) == r['kernel'] and m.group(3) in ('WRITE_SIZE', 'FETCH_SIZE'):
      # (a tuner's candidates launch other grids on the first launches: the grid with the most launches is the settled one)
      key = m.group(3)
      if key not in vals or int(m.group(4)) > vals[key][0]:
        vals[key] = (int(m.group(4)), float(m.group(5)), m.group(2))
  if len(vals) != 2 or len(shapes) != 1:  # (the three profiled runs must have settled on ONE launch shape)
    sys.stderr.write('%s: no record (%s, shapes %s)\n' % (row, sorted(vals), sorted(shapes)))
    continue
  wb, fb = int(round(vals['WRITE_SIZE'][1] * 1024)), int(round(2 * vals['FETCH_SIZE'][1] * 1024))
  records.append({'game': r['game'], 'level': r['level'], 'batch': r['batch'], 'kernel': r['kernel'], 'launch_shape': r['launch_shape'], 'api': 'step',
                  'grid': vals['WRITE_SIZE'][2], 'write_bytes': wb, 'fetch_bytes_corrected': fb, 'bytes_per_launch': wb + fb,
                  'algorithmic_bytes_per_launch': r['algorithmic_bytes_per_env_step'] * r['batch'],
                  'ratio': (wb + fb) / float(r['algorithmic_bytes_per_env_step'] * r['batch']), 'source': source})
print(json.dumps({
    'records': records,
    'source': '%s: WRITE_SIZE (KiB) + 2 x FETCH_SIZE (KiB; gfx950 reports half of a coalesced read stream, MI355X_MICROARCH.md HBM '
              'section), separate --pmc passes with --kernel-trace only, mean over the step launches of the row\'s kernel in the launch '
              'shape the row settled on' % source,
    'command': 'tools/profile_r06.sh (rocprofv3 --kernel-trace --pmc WRITE_SIZE | FETCH_SIZE -- python tools/row_bench.py <row>)'}, indent=1))
