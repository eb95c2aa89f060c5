"""Adapter giving the HIP engine (through pycolab_amd.Engine and the C ABI)
the small surface tests/helpers.replay_trace drives."""
import numpy as np

from pycolab_amd.engine import Engine


class HipAdapter(object):

  def __init__(self, template, batch, auto_reset=True, seed=None, env_offset=0):
    if seed is None:  # the golden traces' RNG seed travels in template.param[0]
      seed = int(template.param[0])
    self.eng = Engine.from_template(template, batch=batch, auto_reset=auto_reset, seed=seed,
                                    env_offset=env_offset)
    self.template = template
    self.batch = batch

  def reset(self):
    self.eng.its_showtime()

  def step(self, actions, auto_reset=True):
    self.eng._auto_reset = bool(auto_reset)
    self.eng.step(np.asarray(actions, np.int32))

  def step_hashed(self, seed, t0, steps, env_offset=0, auto_reset=True):
    self.eng._auto_reset = bool(auto_reset)
    self.eng.step_hashed(seed, t0, steps, env_offset=env_offset)

  def read(self, name):
    if name == 'planes':
      return self.eng.planes_view(host=True)
    return self.eng.buffers[name].numpy()

  def sprites(self):
    st, _ = self.eng._read_things(sprites=True, curtains=False)
    ns = len(self.template.sprites)
    out = np.zeros((self.batch, ns, 5), np.int16)
    for i, f in enumerate(('row', 'col', 'vrow', 'vcol', 'visible')):
      out[:, :, i] = st[f][:, :ns]
    return out

  def curtains(self):
    _, cur = self.eng._read_things(sprites=False, curtains=True)
    return cur[:, :len(self.template.drapes)]
