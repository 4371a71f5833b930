"""Per-phase cycle budget of the 256x256 eight-phase GEMM (lab build: python -m mmt_amd.build --instr;
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so python tools/gemm3_budget.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16
for (M, N, K) in ((8192, 8192, 8192), (8192, 16384, 7168)):
  a = torch.randn(M, K, device=dev).to(bf)
  b = (torch.randn(N, K, device=dev) * 0.05).to(bf)
  out = torch.empty(M, N, device=dev, dtype=torch.float32)
  nb = (M // 256) * (N // 256)
  dbg = torch.zeros(nb * 2, 20, device=dev, dtype=torch.int64)
  for _ in range(2):
    ops.gemm_nt(a, b, out, 'F32', tile=21, seed_dev=dbg)
  torch.cuda.synchronize()
  d = dbg.cpu().double()
  for grp in (0, 1):
    x = d[grp::2]
    x = x[x[:, 17] > 0]
    kt = x[0, 17].item()
    m = x.mean(0)
    print('%d x %d x %d group %d: %d blocks, %d K-tiles, loop %.0f cycles = %.0f per K-tile (MFMA-only: 2048)' % (M, N, K, grp, x.shape[0], kt, m[16], m[16] / kt))
    for p in range(4):
      print('   phase %d: reads + LDS-DMA + vmcnt %5.0f | barrier %5.0f | lgkm wait + 8 MFMA %5.0f | barrier %5.0f' %
            (p + 1, m[p * 4] / kt, m[p * 4 + 1] / kt, m[p * 4 + 2] / kt, m[p * 4 + 3] / kt))
