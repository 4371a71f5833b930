"""Per-phase cycle budget of the 256x256 weight-gradient kernel (wgrad3.hip) on configs[4]'s shapes.  Lab build:
   python -m mmt_amd.build --instr;  MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so python tools/wgrad3_budget.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
rows, live, d, inter = 14464, 14393, 1024, 6144
items = []
for (N, K2) in [(inter, d), (d, inter), (3 * d, d), (d, d)]:
  a = (torch.randn(rows, N, device=dev) * 0.5).to(torch.bfloat16)
  b = (torch.randn(rows, K2, device=dev) * 0.5).to(torch.bfloat16)
  items.append((a, b, torch.empty(N, K2, device=dev), torch.empty(N, device=dev)))
nrd = torch.tensor([live], device=dev, dtype=torch.int32)
dbg = torch.zeros(256 * 2, 20, device=dev, dtype=torch.int64)
L = ctypes.CDLL(_lib.LIB_PATH)
L.mmt_debug_set_wgrad3_buffer(ctypes.c_void_p(dbg.data_ptr()))
for _ in range(2):
  ops.wgrad_grouped(items, rows, n_rows_dev=nrd)
torch.cuda.synchronize()
x = dbg.cpu().double()
for grp in (0, 1):
  for bias in (0, 1):
    y = x[grp::2]
    y = y[(y[:, 17] > 0) & (y[:, 18] == bias)]
    if not len(y):
      continue
    kt = y[0, 17].item()
    m = y.mean(0)
    print('group %d, %s blocks (%d): %d units, loop %.0f cycles = %.0f per unit (MFMA-only: 2048)' %
          (grp, 'bias-summing' if bias else 'plain', len(y), kt, m[16], m[16] / kt))
    for p in range(4):
      print('   phase %d: reads + requests + vmcnt %5.0f | barrier %5.0f | lgkm wait + 16 MFMA %5.0f | barrier %5.0f' %
            (p + 1, m[p * 4] / kt, m[p * 4 + 1] / kt, m[p * 4 + 2] / kt, m[p * 4 + 3] / kt))
