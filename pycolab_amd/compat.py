"""Run game files written for `pycolab` against this package, unchanged.

`aliased()` temporarily makes `import pycolab...` resolve to `pycolab_amd...`,
so a reference example such as `examples/scrolly_maze.py` can be loaded by
path and its `make_game()` builds a `pycolab_amd.engine.Engine`.
"""

import contextlib
import importlib
import importlib.util
import sys
import types

_SUBMODULES = ['ascii_art', 'engine', 'things', 'plot', 'rendering', 'human_ui',
               'cropping', 'storytelling', 'prefab_parts', 'prefab_parts.sprites',
               'prefab_parts.drapes', 'protocols', 'protocols.scrolling',
               'protocols.logging']


@contextlib.contextmanager
def aliased():
  saved = {k: v for k, v in sys.modules.items()
           if k == 'pycolab' or k.startswith('pycolab.')}
  for k in saved:
    del sys.modules[k]
  try:
    root = importlib.import_module('pycolab_amd')
    sys.modules['pycolab'] = root
    for name in _SUBMODULES:
      try:
        sys.modules['pycolab.' + name] = importlib.import_module('pycolab_amd.' + name)
      except ImportError:
        pass
    # `import curses` at the top of example files must not fail headless.
    if 'curses' not in sys.modules:
      try:
        importlib.import_module('curses')
      except ImportError:
        sys.modules['curses'] = types.ModuleType('curses')
    yield
  finally:
    for k in [k for k in sys.modules if k == 'pycolab' or k.startswith('pycolab.')]:
      del sys.modules[k]
    sys.modules.update(saved)


def load_game_module(path, name=None):
  """Import the game file at `path` with `pycolab` aliased to this package."""
  name = name or ('pcx_game_' + path.replace('/', '_').replace('.', '_'))
  with aliased():
    spec = importlib.util.spec_from_file_location(name, path)
    module = importlib.util.module_from_spec(spec)
    sys.modules[name] = module  # inspect.getsource(cls) finds a class through its module
    spec.loader.exec_module(module)
  return module
