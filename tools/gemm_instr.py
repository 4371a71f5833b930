"""Where do the cycles of gemm2's K-loop go?  Needs the lab build:  python -m mmt_amd.build --instr
   MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so python tools/gemm_instr.py
Per block, wave 0 accumulates s_memtime deltas for: counted-vmcnt wait, barrier, LDS-DMA issue, LDS-read+MFMA."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def run(rows, N, K, tile, epi='F32'):
  R = ops.pad_rows(rows)
  a = torch.randn(R, K, device=dev).to(torch.bfloat16)
  b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
  out = torch.zeros(R, N, device=dev, dtype=torch.float32 if epi == 'F32' else torch.bfloat16)
  nblk = 8192
  dbg = torch.zeros(nblk, 16, device=dev, dtype=torch.int64)
  for _ in range(3):
    ops.gemm_nt(a, b, out, epi, m=rows, tile=tile, seed_dev=dbg)
  torch.cuda.synchronize()
  d = dbg.cpu().double()
  d = d[d[:, 7] > 0]
  kt = d[0, 7].item()
  m = d.mean(0)
  span = d[:, 6].mean().item()  # whole block (shader cycles)
  print('rows %5d N %4d K %4d tile %2d blocks %4d KT %3d | per K-step cycles: wait %6.0f barrier %6.0f issue %6.0f compute %6.0f '
        '| loop %8.0f epilogue %7.0f | whole block %8.0f' %
        (rows, N, K, tile, d.shape[0], kt, m[0] / kt, m[1] / kt, m[2] / kt, m[3] / kt, m[4], m[5], span))


tiles = [int(t) for t in sys.argv[1:]] or [5, 7, 3, 11, 12, 13, 18]
for rows in (3596, 6976):
  for (N, K) in ((512, 3072), (512, 512), (3072, 512)):
    for tile in tiles:
      run(rows, N, K, tile)
