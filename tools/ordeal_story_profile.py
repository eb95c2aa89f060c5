#!/usr/bin/env python3
"""cProfile of the batched ordeal Story's host side (tools/ordeal_story_bench.py main, 65,536 environments x 160 steps)."""
import cProfile
import importlib.util
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ['x', '--batch', '65536', '--steps', '160']
spec = importlib.util.spec_from_file_location('ob', os.path.join(ROOT, 'tools', 'ordeal_story_bench.py'))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
m.cpu_reference = lambda tr, seconds=0: None
cProfile.run('m.main()', '/tmp/prof.out')
pstats.Stats('/tmp/prof.out').sort_stats('cumulative').print_stats(45)
