"""The 256x256 eight-phase GEMM (gemm3.hip, tile 21) on large shapes: TFLOP/s against the 128x128 tile (14) and the vendor
library, random operands.   python tools/gemm3_lab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402
from tools.gemm_lab import timeit  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16
for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (8192, 65536, 7168), (14464, 6144, 1024), (14464, 1024, 6144)):
  a = (torch.randn(M, K, device=dev)).to(bf)
  b = (torch.randn(N, K, device=dev) * 0.05).to(bf)
  out = torch.empty(M, N, device=dev, dtype=torch.float32 if M * N <= 2 ** 29 else bf)
  epi = 'F32' if out.dtype == torch.float32 else 'BF16'
  ref = None
  if M * N <= 2 ** 27:
    ref = a.float() @ b.float().t()
  res = []
  for tile in (14, 21):
    ops.gemm_nt(a, b, out, epi, tile=tile)
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item() if ref is not None else float('nan')
    res.append((tile, err))
  fns = [lambda t=t: ops.gemm_nt(a, b, out, epi, tile=t) for t in (14, 21)]
  o16 = torch.empty(M, N, device=dev, dtype=bf)
  fns.append(lambda: torch.matmul(a, b.t(), out=o16))
  ts = timeit(fns, iters=5 if M * N * K > 2 ** 38 else 20)
  fl = 2.0 * M * N * K
  print('%6d x %6d x %5d %s | tile 14: %8.1f us %6.0f TF (err %.1e) | tile 21: %8.1f us %6.0f TF = %.3f of 2.5 PF (err %.1e) | vendor bf16: %8.1f us %6.0f TF'
        % (M, N, K, epi, ts[0], fl / ts[0] / 1e6, res[0][1], ts[1], fl / ts[1] / 1e6, fl / ts[1] / 1e6 / 2500, res[1][1], ts[2],
           fl / ts[2] / 1e6))
  del a, b, out, o16, ref
  torch.cuda.empty_cache()
