"""Fused Adam throughput on the two flat-buffer sizes of the step (17.1 M video side, 108 M text tower)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import _lib, ops  # noqa: E402
from tools.gemm_lab import timeit  # noqa: E402

L = _lib.lib()
for n in (17_100_000 // 64 * 64, 108_300_000 // 64 * 64):
  p, g, m, v = (torch.randn(n, device='cuda') * 0.01 for _ in range(4))
  v.abs_()
  step = torch.ones(1, dtype=torch.int32, device='cuda')
  fn = lambda: _lib.check(L.mmt_adam_step(ops._p(p), ops._p(g), ops._p(m), ops._p(v), n, 1e-4, 0.9, 0.999, 1e-8, 0.0,
                                          ops._p(step), None, ops._stream()), 'adam')
  torch.cuda.synchronize()
  t, = timeit([fn], iters=5)
  print('adam n = %11d : %7.1f us  %5.2f TB/s' % (n, t, n * 28 / t / 1e6))
