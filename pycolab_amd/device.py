"""Device memory and stream plumbing for the host facade.

PyTorch-ROCm, when present with a visible GPU, owns the allocations (so
observations are ordinary device tensors for the consumer) and supplies the
HIP stream; otherwise buffers come from the C ABI's own allocator and are
read back with synchronous copies.  Either way the step kernels are the ones
in csrc/ -- nothing here computes.
"""

import ctypes

import numpy as np

from pycolab_amd import _native as N

from pycolab_amd._torchprobe import torch_module  # noqa: F401 (re-exported)


_NP_TO_TORCH = {'uint8': 'uint8', 'int32': 'int32', 'float32': 'float32', 'int64': 'int64', 'float64': 'float64'}


class DeviceBuffer(object):
  """A typed device array with `.ptr`, `.numpy()` and (with torch) `.tensor`."""

  def __init__(self, shape, dtype, device_id):
    self.shape = tuple(int(s) for s in shape)
    self.dtype = np.dtype(dtype)
    self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
    self.device_id = device_id
    torch = torch_module()
    self.tensor = None
    self._raw = None
    if torch is not None:
      self.tensor = torch.zeros(self.shape, dtype=getattr(torch, _NP_TO_TORCH[self.dtype.name]),
                                device='cuda:%d' % device_id)
      self.ptr = self.tensor.data_ptr()
    else:
      p = ctypes.c_void_p()
      N.check(N.lib().pcx_device_malloc(ctypes.byref(p), max(self.nbytes, 1)))
      self._raw = p
      self.ptr = p.value
      self.upload(np.zeros(self.shape, self.dtype))

  @classmethod
  def view_of(cls, tensor, dtype, device_id):
    """A typed buffer over (a slice of) an existing device tensor: no allocation."""
    self = cls.__new__(cls)
    self.dtype = np.dtype(dtype)
    torch = torch_module()
    self.tensor = tensor.view(getattr(torch, _NP_TO_TORCH[self.dtype.name]))
    self.shape = tuple(self.tensor.shape)
    self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
    self.device_id = device_id
    self._raw = None
    self.ptr = self.tensor.data_ptr()
    return self

  def numpy(self):
    if self.tensor is not None:
      return self.tensor.cpu().numpy()
    out = np.empty(self.shape, self.dtype)
    N.check(N.lib().pcx_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes))
    return out

  def upload(self, array):
    array = np.ascontiguousarray(array, dtype=self.dtype).reshape(self.shape)
    if self.tensor is not None:
      torch = torch_module()
      self.tensor.copy_(torch.from_numpy(array))
    else:
      N.check(N.lib().pcx_memcpy_h2d(self.ptr, array.ctypes.data, self.nbytes))

  def free(self):
    if self._raw is not None:
      N.lib().pcx_device_free(self._raw)
      self._raw = None
    self.tensor = None

  def __del__(self):
    try:
      self.free()
    except Exception:  # pylint: disable=broad-except
      pass


def current_stream(device_id):
  torch = torch_module()
  if torch is None:
    return None
  return ctypes.c_void_p(torch.cuda.current_stream(device_id).cuda_stream)


def synchronize(device_id):
  torch = torch_module()
  if torch is not None:
    torch.cuda.synchronize(device_id)
  else:
    N.check(N.lib().pcx_stream_synchronize(None))
