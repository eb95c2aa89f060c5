"""Observation containers (reference: rendering.py:28-63).

`Observation` is the reference's namedtuple.  In batched mode `board` has
shape [B, rows, cols] and every `layers[c]` has shape [B, rows, cols]; all of
them are zero-copy views of the engine's `planes` array
([B, 1 + n_chars, rows, cols] uint8 in HBM), valid until the next step --
the same aliasing rule as the reference.
"""

import collections

Observation = collections.namedtuple('Observation', ['board', 'layers'])
