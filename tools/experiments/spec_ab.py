#!/usr/bin/env python3
"""Digest of what the table-driven kernel leaves after T hashed steps of a fixture: run once with the stock library and
once with a build specialised for the template (PCX_LIB=...), the digests must agree.
  python tools/experiments/spec_ab.py warehouse_L0 4099 200"""
import hashlib, os, sys
os.environ.setdefault('PCX_FORCE_GENERIC', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pycolab_amd.compiler import GameTemplate
from pycolab_amd.engine import Engine

name, batch, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = GameTemplate.load(os.path.join(root, 'tests', 'golden', 'templates', name + '.npz'))
eng = Engine.from_template(t, batch=batch, auto_reset=True, seed=7)
eng.its_showtime()
h = hashlib.sha256()
for t0 in range(0, T, 8):
  eng.step_hashed(0xABCDEF, t0, 8)
  h.update(eng.planes_view().cpu().numpy().tobytes())
  for nm in ('reward', 'reward_set', 'discount', 'done', 'frame', 'error'):
    h.update(eng.buffers[nm].tensor.cpu().numpy().tobytes())
print(name, batch, T, h.hexdigest()[:24], 'errors', int(eng.buffers['error'].tensor.sum()))
