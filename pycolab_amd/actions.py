"""Synthetic action tapes shared by host, oracle and device.

`pcx_action_hash` (include/pcx.h) is the counter-based generator behind
`Engine.step_hashed`: action of environment `env` (its GLOBAL index in a
sharded batch) at step `t` = hash(seed, env, t) % n_actions.  This is its
vectorised host twin, so that a sharded run can stage on every rank exactly the
tape an unsharded engine would draw on the device (the global
environment index drives the actions, never the rank).
"""

import numpy as np

_M1, _M2, _M3 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)


def action_hash(seed, env, t):
  """uint32 array: pcx_action_hash(seed, env, t), broadcasting `env` and `t`."""
  with np.errstate(over='ignore'):
    env = np.asarray(env, dtype=np.uint64)
    t = np.asarray(t, dtype=np.uint64)
    x = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (env * _M1) ^ (t * _M2)
    x = (x ^ (x >> np.uint64(30))) * _M2
    x = (x ^ (x >> np.uint64(27))) * _M3
    x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(32)).astype(np.uint32)


def hashed_tape(seed, env_lo, n_envs, t0, steps, n_actions):
  """int32 [steps, n_envs]: the actions `step_hashed(seed, t0, steps)` draws for
  global environments [env_lo, env_lo + n_envs)."""
  env = np.arange(env_lo, env_lo + n_envs, dtype=np.uint64)[None, :]
  t = np.arange(t0, t0 + steps, dtype=np.uint64)[:, None]
  return (action_hash(seed, env, t) % np.uint32(n_actions)).astype(np.int32)
