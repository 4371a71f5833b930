"""Workgroup dispatch rate on MI355X as a function of the workgroup shape (threads, LDS bytes) and the grid size: empty
blocks (spin = 0) and blocks that stay resident for ~2 us (spin = 200 ticks of the 100 MHz clock64 counter)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import _lib, ops  # noqa: E402
from tools.gemm_lab import timeit  # noqa: E402

L = _lib.lib()
sink = torch.zeros(4, device='cuda')
for threads, lds in ((256, 1024), (256, 16 * 1024), (256, 33 * 1024), (256, 66 * 1024), (512, 66 * 1024), (256, 130 * 1024),
                     (512, 130 * 1024), (1024, 8 * 1024)):
  for spin in (0, 200):
    row = []
    for blocks in (256, 512, 1024, 2048, 4096):
      fn = lambda b=blocks: _lib.check(L.mmt_debug_dispatch_probe(b, threads, lds, spin, ops._p(sink), ops._stream()), 'p')
      torch.cuda.synchronize()
      t, = timeit([fn], iters=10)
      row.append('%5d: %6.1f us' % (blocks, t))
    print('threads %4d lds %6d B spin %3d | %s' % (threads, lds, spin, '  '.join(row)))
