#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a PMC summary (tools/pmc_summary.py output):
HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE (both in KiB; gfx950 reports
half of a coalesced read stream -- MI355X_MICROARCH.md, HBM section), per kernel
and launch size of `bench.py`'s default run.

  python tools/traffic_records.py profiles/r02_pmc_summary.txt > profiles/hbm_traffic.json
"""
import json
import re
import sys

# (kernel, rocprofv3 Grid_Size) -> (game, level, batch) of bench.py's headline + other_configs launches
LAUNCHES = {
    ('pcx_scrolly_maze_step', '1048576'): ('scrolly_maze', 0, 1048576),
    ('pcx_scrolly_maze_step', '131072'): ('scrolly_maze', 0, 4096),        # round 3: 256 workgroups of 16 environments x 8 cooperating waves
    ('pcx_marauders_step', '131072'): ('marauders', 0, 32768),             # 512 groups x 4 waves
    ('pcx_warehouse_step', '262144'): ('warehouse', 0, 262144),
    ('pcx_better_scrolly_step', '65536'): ('better_scrolly_maze', 0, 65536),
    ('pcx_hello_world_step', '1048576'): ('hello_world', 0, 1048576),
}
vals = {}
for line in open(sys.argv[1]):
  m = re.match(r'(\S+)\s+grid (\S+)\s+(\S+)\s+n=\s*\d+ mean=\s*([\d.]+)', line)
  if m:
    vals[(m.group(1), m.group(2), m.group(3))] = float(m.group(4))
records = []
for (kernel, grid), (game, level, batch) in LAUNCHES.items():
  w, f = vals.get((kernel, grid, 'WRITE_SIZE')), vals.get((kernel, grid, 'FETCH_SIZE'))
  if w is None or f is None:
    continue
  wb, fb = int(round(w * 1024)), int(round(2 * f * 1024))
  records.append({'game': game, 'batch': batch, 'level': level, 'kernel': kernel, 'write_bytes': wb,
                  'fetch_bytes_corrected': fb, 'bytes_per_launch': wb + fb})
print(json.dumps({
    'records': records,
    'source': '%s: WRITE_SIZE (KiB) + 2 x FETCH_SIZE (KiB; gfx950 reports half of a coalesced read stream, '
              'MI355X_MICROARCH.md HBM section), separate --pmc passes with --kernel-trace only, mean over the step '
              'launches of each kernel' % sys.argv[1],
    'command': 'tools/profile_r02.sh <tag> (rocprofv3 --kernel-trace --pmc WRITE_SIZE | FETCH_SIZE -- python bench.py '
               '--steps 20 --warmup 3 --no-cpu-baseline)'}, indent=1))
