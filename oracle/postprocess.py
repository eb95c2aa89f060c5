"""TEST INFRASTRUCTURE: numpy restatement of the reference's observation
post-processors (pycolab/rendering.py:304-661), one environment at a time.
Pinned by tests/test_postprocess.py against outputs of the reference's own
classes recorded in tests/golden/traces (oracle/gen_golden.py)."""
import numpy as np


def to_array(board, value_mapping, dtype=None, permute=None):
  """rendering.ObservationToArray.__call__ (:484-542)."""
  first = next(iter(value_mapping.values()))
  dtype = np.dtype(dtype) if dtype is not None else np.array(first).dtype
  try:
    depth, is_3d = len(first), True
  except TypeError:
    depth, is_3d = 1, False
  out = np.zeros((depth,) + board.shape, dtype)
  for ascii_value in np.unique(board):
    if chr(ascii_value) not in value_mapping:
      raise RuntimeError('unmapped character %r' % chr(ascii_value))
    value = value_mapping[chr(ascii_value)]
    mask = board == ascii_value
    if is_3d:
      for layer, component in enumerate(value):
        out[layer, mask] = component
    else:
      out[:, mask] = value
  result = out if is_3d else out[0]
  return result if permute is None else np.transpose(result, permute)


def feature_array(layers, chars, shape, permute=None):
  """rendering.ObservationToFeatureArray.__call__ (:610-661)."""
  out = np.zeros((len(chars),) + tuple(shape), np.float32)
  for i, c in enumerate(chars):
    if c in layers:
      out[i] = layers[c]
  return out if permute is None else np.transpose(out, permute)


def repaint(board, layer_chars, mapping):
  """rendering.ObservationCharacterRepainter.__call__ (:340-406)."""
  lut = np.arange(128, dtype=np.uint8)
  for k, v in mapping.items():
    lut[ord(k)] = ord(v)
  new_board = lut[board]
  out_chars = sorted((set(layer_chars) - set(mapping)).union(mapping.values()))
  return new_board, {c: new_board == ord(c) for c in out_chars}
