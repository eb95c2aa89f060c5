#!/usr/bin/env python3
"""What one dependent kernel launch costs on this box: a near-empty kernel (pcx_device_fill_probe over 1 KiB),
launched back to back on one stream, HIP events."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pycolab_amd import _native as N
buf = torch.empty(1 << 20, dtype=torch.uint8, device='cuda')
stream = ctypes.c_void_p(torch.cuda.current_stream(0).cuda_stream)
lib = N.lib()
for n in (2000, 2000):
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    lib.pcx_device_fill_probe(buf.data_ptr(), 1024, stream)
  e1.record(); torch.cuda.synchronize()
  print('near-empty kernel, %d launches back to back: %.2f us per launch' % (n, e0.elapsed_time(e1) / n * 1e3))
