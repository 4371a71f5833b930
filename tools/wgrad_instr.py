"""Cycle breakdown of wgrad_grouped_kernel's K-loop (lab build: python -m mmt_amd.build --instr;
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so python tools/wgrad_instr.py)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
L = ctypes.CDLL(_lib.LIB_PATH)
for rows in (1750, 3596, 6976):
  R = ops.pad_rows(rows)
  d, I = 512, 3072
  bf = torch.bfloat16
  rnd = lambda *s: torch.randn(*s, device=dev).to(bf)
  items = [(rnd(R, I), rnd(R, d), torch.zeros(I, d, device=dev), torch.zeros(I, device=dev)),
           (rnd(R, d), rnd(R, I), torch.zeros(d, I, device=dev), torch.zeros(d, device=dev)),
           (rnd(R, 3 * d), rnd(R, d), torch.zeros(3 * d, d, device=dev), torch.zeros(3 * d, device=dev)),
           (rnd(R, d), rnd(R, d), torch.zeros(d, d, device=dev), torch.zeros(d, device=dev))]
  dbg = torch.zeros(1024, 8, device=dev, dtype=torch.int64)
  L.mmt_debug_set_wgrad_buffer(ctypes.c_void_p(dbg.data_ptr()))
  for _ in range(3):
    ops.wgrad_grouped(items, rows)
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(10):
    ops.wgrad_grouped(items, rows)
  e.record()
  torch.cuda.synchronize()
  x = dbg.cpu().double()
  x = x[x[:, 5] > 0]
  st = x[0, 5].item()
  m = x.mean(0)
  print('rows %d: %.1f us/launch, %d blocks, %d steps | per step cycles: wait %.0f barrier %.0f issue %.0f compute %.0f | loop %.0f cycles'
        % (rows, s.elapsed_time(e) * 100, x.shape[0], st, m[0] / st, m[1] / st, m[2] / st, m[3] / st, m[4]))
