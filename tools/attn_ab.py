"""Same-box A/B of two builds of the attention forward kernel: libmmt_hip.so (A) vs libmmt_hip_instr.so (B)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import _lib, ops  # noqa: E402
from mmt_amd.ops import _p, _stream  # noqa: E402
from tools.gemm_lab import timeit  # noqa: E402

A = _lib.lib()
B = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), 'libmmt_hip_instr.so'))
B.mmt_attn_fwd.argtypes = A.mmt_attn_fwd.argtypes
B.mmt_attn_fwd.restype = ctypes.c_int
dev = torch.device('cuda:0')
H, d = 4, 512
for Bz, S in ((32, 218), (32, 128), (32, 64), (128, 218)):
  rows = Bz * S
  R = ops.pad_rows(rows)
  qkv = (torch.randn(R, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
  mask = torch.zeros(R, device=dev)
  cu = torch.arange(0, Bz + 1, device=dev, dtype=torch.int32) * S
  ctxa, ctxb = (torch.zeros(R, d, device=dev, dtype=torch.bfloat16) for _ in range(2))
  lse = torch.zeros(R, H, device=dev)
  thr, sc = ops.dropout_params(0.1)
  fa = lambda: _lib.check(A.mmt_attn_fwd(_p(qkv), _p(cu), _p(mask), _p(ctxa), _p(lse), Bz, S, H, d, 128 ** -0.5, 7, thr, sc, None, _stream()), 'a')
  fb = lambda: _lib.check(B.mmt_attn_fwd(_p(qkv), _p(cu), _p(mask), _p(ctxb), _p(lse), Bz, S, H, d, 128 ** -0.5, 7, thr, sc, None, _stream()), 'b')
  torch.cuda.synchronize()
  ta, tb = timeit([fa, fb], rounds=5)
  err = (ctxa[:rows].float() - ctxb[:rows].float()).abs().max().item()
  print('B %4d S %4d | A %6.2f us  B %6.2f us  (max |A-B| %.1e)' % (Bz, S, ta, tb, err))
