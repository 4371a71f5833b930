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
  dbg = torch.zeros(2048, 8, device=dev, dtype=torch.int64)
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
  if os.environ.get('MMT_WGRAD_LOCKSTEP', '0') != '1':
    # phased loop: rows 2b / 2b+1 = wave 0 of group 0 / 1 of block b; [5] = units (half-steps) of the block
    g0, g1 = x[0::2], x[1::2]
    ok = g0[:, 5] > 0
    print('rows %d: prologue (entry -> loop) %.0f cycles, loop %.0f, epilogue (reduction + stores) %.0f; block start spread %.0f cycles, '
          'last block ends %.0f cycles after the first starts'
          % (rows, g0[ok, 6].mean(), g0[ok, 4].mean(), g1[ok, 7].mean(), (g0[ok, 7].max() - g0[ok, 7].min()),
             (g0[ok, 7] + g0[ok, 6] + g0[ok, 4] + g1[ok, 7]).max() - g0[ok, 7].min()))
    for kg in (0, 1):
      y = x[kg::2]
      y = y[y[:, 5] > 0]
      u = y[0, 5].item()
      m = y.mean(0)
      print('rows %d group %d: %.1f us/launch, %d blocks, %d half-steps | per half-step cycles: vmcnt wait %.0f barrier %.0f '
            'issue %.0f compute %.0f | loop %.0f cycles = %.0f per half-step'
            % (rows, kg, s.elapsed_time(e) * 100, y.shape[0], u, m[0] / u, m[1] / u, m[2] / u, m[3] / u, m[4], m[4] / u))
    continue
  x = x[x[:, 5] > 0]
  st = x[0, 5].item()
  m = x.mean(0)
  print('rows %d: %.1f us/launch, %d blocks, %d steps | per step cycles: wait %.0f barrier %.0f issue %.0f compute %.0f | loop %.0f cycles'
        % (rows, s.elapsed_time(e) * 100, x.shape[0], st, m[0] / st, m[1] / st, m[2] / st, m[3] / st, m[4]))
