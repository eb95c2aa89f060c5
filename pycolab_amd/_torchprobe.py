"""Locate PyTorch-ROCm (optional plumbing: device memory, streams, RCCL).

Imported *before* csrc/libpcx.so is loaded: torch ships its own copy of the
HIP runtime, and whichever copy is loaded first is the one the process uses;
loading libpcx.so first would leave torch with a second, device-less runtime.
"""

import os

_torch = None


def torch_module():
  """torch if importable with a usable GPU (and not disabled), else None."""
  global _torch
  if _torch is None:
    _torch = False
    if os.environ.get('PCX_NO_TORCH', '0') != '1':
      try:
        import torch
        if torch.cuda.is_available():
          _torch = torch
      except ImportError:
        pass
  return _torch or None
