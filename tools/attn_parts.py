"""What is the attention forward kernel's time made of?  Lab build (python -m mmt_amd.build --instr; run with
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so): parts of the kernel are switched off through mmt_debug_set_att_lab."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import _lib, ops  # noqa: E402
from mmt_amd.ops import _p, _stream  # noqa: E402
from tools.gemm_lab import timeit  # noqa: E402

import ctypes
L = ctypes.CDLL(os.environ['MMT_HIP_LIB'])
L.mmt_debug_set_att_lab.argtypes = [ctypes.c_int]
dev = torch.device('cuda:0')
H, d = 4, 512
lib = _lib.lib()
for B, S in ((32, 218), (32, 128)):
  rows = B * S
  R = ops.pad_rows(rows)
  qkv = (torch.randn(R, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
  mask = torch.zeros(R, device=dev)
  cu = torch.arange(0, B + 1, device=dev, dtype=torch.int32) * S
  ctx = torch.zeros(R, d, device=dev, dtype=torch.bfloat16)
  lse = torch.zeros(R, H, device=dev)
  thr, sc = ops.dropout_params(0.1)
  fwd = lambda: _lib.check(lib.mmt_attn_fwd(_p(qkv), _p(cu), _p(mask), _p(ctx), _p(lse), B, S, H, d, 128 ** -0.5, 7, thr, sc,
                                            None, _stream()), 'f')
  out = []
  for bits, name in ((0, 'all'), (1, 'no prefetch DMA'), (2, 'no QK'), (8, 'no PV'), (16, 'QK only'), (1 + 2 + 16, 'loop = barriers only'),
                     (1 + 2 + 16 + 32, '+ no first DMA'), (1 + 2 + 16 + 32 + 64, '+ no stores')):
    L.mmt_debug_set_att_lab(bits)
    torch.cuda.synchronize()
    t, = timeit([fwd])
    out.append('%s %.1f' % (name, t))
  print('B %4d S %4d | %s' % (B, S, '  '.join(out)))
