#!/usr/bin/env python3
"""profiles/r06_kernel_stats.csv and profiles/r06_pmc_summary.txt from tools/profile_r06.sh's per-row output
(gpurun_out/prof_r06/<row>/{row.json,kernel_stats.csv,pmc_summary.txt}): the step kernels' lines of every row's rocprofv3
kernel stats next to the HIP-event average of the same run, and the rows' PMC summaries one after the other.
  python tools/aggregate_prof_r06.py gpurun_out/prof_r06 profiles"""
import json
import os
import sys

root, out = sys.argv[1], sys.argv[2]
ORDER = ['headline', 'scrolly_131072', 'scrolly_262144', 'scrolly_custom_H_131072', 'scrolly_L1_131072', 'scrolly_4096', 'marauders_32768',
         'marauders_262144', 'warehouse_262144', 'better_scrolly_65536', 'hello_world_1048576', 'marauders_custom_A', 'walkers',
         'warehouse_generic', 'ordeal_kansas']
rows = [r for r in ORDER if os.path.isdir(os.path.join(root, r))] + sorted(set(os.listdir(root)) - set(ORDER))
stats = ['# rocprofv3 --kernel-trace --stats per row of bench.py\'s line (tools/profile_r06.sh: `python tools/row_bench.py <row>` = bench.measure_config of that row, tuner settled before timing);',
         '# the step kernel\'s line of each row\'s kernel_stats.csv: "Name","Calls","TotalDurationNs","AverageNs",... -- next to the HIP-event average the same run printed (row.json: ms_per_step).',
         '# The whole of `python bench.py` under rocprofv3 (all rows in one trace): r06_bench_under_rocprof_kernel_stats.csv + r06_bench_under_rocprof.json.']
pmc = ['# separate rocprofv3 --kernel-trace --pmc passes per row (WRITE_SIZE, FETCH_SIZE; the headline row also SQ counters), tools/pmc_summary.py: mean per launch by grid (KiB for the two sizes).',
       '# HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE (gfx950 reports half of a coalesced read stream: MI355X_MICROARCH.md, HBM section) -> profiles/hbm_traffic.json (tools/traffic_records_r06.py)']
for row in rows:
  d = os.path.join(root, row)
  try:
    r = json.load(open(os.path.join(d, 'row.json')))
  except Exception:  # pylint: disable=broad-except
    sys.stderr.write('%s: no row.json\n' % row)
    continue
  head = '== %s: %s, launch shape %s' % (row, r['workload'], r['launch_shape'])
  stats.append('%s, HIP events %.4f ms per step (%.3f of 8 TB/s)' % (head, r['ms_per_step'], r['hbm_frac']))
  lines = open(os.path.join(d, 'kernel_stats.csv')).read().splitlines()
  stats.append(lines[0])
  stats += [l for l in lines[1:] if '_step' in l.split(',')[0] or 'pcx_' in l.split('(')[0]][:3]
  pmc.append(head)
  pmc += [l for l in open(os.path.join(d, 'pmc_summary.txt')).read().splitlines() if l.strip()]
open(os.path.join(out, 'r06_kernel_stats.csv'), 'w').write('\n'.join(stats) + '\n')
open(os.path.join(out, 'r06_pmc_summary.txt'), 'w').write('\n'.join(pmc) + '\n')
print('%d rows' % len(rows))
