#!/usr/bin/env python3
"""One row of bench.py's line by itself (what tools/profile_r06.sh runs under rocprofv3, so that the only step kernel in a
trace is the row's own): prints the row's JSON.
  python tools/row_bench.py headline | scrolly_131072 | scrolly_262144 | scrolly_custom_H_131072 | scrolly_L1_131072 | scrolly_4096 | marauders_32768 | marauders_262144 |
                            warehouse_262144 | better_scrolly_65536 | hello_world_1048576 | marauders_custom_A | walkers | warehouse_generic | ordeal_kansas"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ROWS = {  # name -> measure_config arguments (game, level, batch, steps, warmup) + keywords
    'headline': ('scrolly_maze', 0, 1048576, 100, 10, {}),
    'scrolly_131072': ('scrolly_maze', 0, 131072, 200, 20, {}),
    'scrolly_262144': ('scrolly_maze', 0, 262144, 200, 20, {}),
    'scrolly_custom_H_131072': ('scrolly_custom_H', 0, 131072, 200, 20, {}),
    'scrolly_L1_131072': ('scrolly_maze', 1, 131072, 200, 20, {}),
    'scrolly_4096': ('scrolly_maze', 0, 4096, 200, 20, {}),
    'marauders_32768': ('marauders', 0, 32768, 200, 20, {}),
    'marauders_262144': ('marauders', 0, 262144, 50, 10, {}),
    'warehouse_262144': ('warehouse', 0, 262144, 100, 10, {}),
    'better_scrolly_65536': ('better_scrolly_maze', 0, 65536, 50, 10, {}),
    'hello_world_1048576': ('hello_world', 0, 1048576, 50, 10, {}),
    'marauders_custom_A': ('marauders_custom_A', 0, 32768, 200, 30, {}),
    'walkers': ('walkers_scroll_groups', 0, 262144, 100, 30, {'cardinal_fields': 2}),
    'warehouse_generic': ('warehouse_generic', 0, 262144, 100, 30, {}),
    'ordeal_kansas': ('ordeal_kansas', 0, 262144, 100, 30, {}),
}


def main():
  name = sys.argv[1]
  game, level, batch, steps, warmup, kw = ROWS[name]
  row = bench.measure_config(game, level, batch, steps, warmup, 0, **kw)
  row['row'] = name
  row['game'], row['level'], row['batch'] = game, level, batch
  print(json.dumps(row))


if __name__ == '__main__':
  main()
