import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import binding
from tests import helpers
from tests.hip_adapter import HipAdapter
shape = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 200
grid = sys.argv[3] if len(sys.argv) > 3 else '1'
os.environ.update(PCX_COOP_BELOW='0', PCX_SM_SHAPE=str(shape), PCX_SM_GRID=grid)
t = helpers.load_template('scrolly_maze_L0')
hip = HipAdapter(t, B)
orc = binding.OracleEngine(t, B)
hip.reset(); orc.reset()
from pycolab_amd import _native as N
for step in range(6):
  hip.step_hashed(0x5EED, step, 1); orc.step_hashed(0x5EED, step, 1)
  print('step', step, 'shape', N.lib().pcx_engine_launch_shape(hip.eng._native))
  for name in ('frame', 'reward', 'done', 'error', 'discount'):
    g, w = hip.read(name), np.array(getattr(orc, name))
    bad = np.nonzero(g != w)[0]
    if len(bad): print('  ', name, 'differs in', len(bad), 'envs; first', bad[:10], 'got', g[bad[:5]], 'want', w[bad[:5]])
  g, w = hip.read('planes'), np.array(orc.planes)
  bad = np.nonzero((g != w).reshape(B, -1).any(axis=1))[0]
  print('   planes differ in', len(bad), 'envs:', bad[:20])
  sp, so = hip.sprites(), orc.sprites()
  bad = np.nonzero((sp != so).reshape(B, -1).any(axis=1))[0]
  print('   sprites differ in', len(bad), 'envs:', bad[:20])
  if len(bad): print(sp[bad[0]], so[bad[0]])
