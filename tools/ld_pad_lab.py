"""Does the leading dimension of the GEMM operands matter?  A K-tile of a row-major [rows, K] operand is one 128-byte segment per
row, rows lda * 2 bytes apart; with lda * 2 a multiple of 1 KiB .. 16 KiB every row of every tile of every CU asks the same
few L2 channels at the same time.  Same GEMMs with the operands' leading dimension padded by `pad` elements.
   python tools/ld_pad_lab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402
from tools.gemm_lab import timeit  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16
cases = [(3573, 3072, 512, 14, 'BF16'), (3573, 512, 3072, 18, 'F32'), (3573, 512, 3072, 13, 'F32'), (3573, 1536, 512, 14, 'BF16'),
         (8192, 8192, 8192, 21, 'F32'), (14464, 1024, 6144, 21, 'F32'), (14464, 6144, 1024, 21, 'F32')]
for (M, N, K, tile, epi) in cases:
  R = ops.pad_rows(M)
  res = []
  for pad in (0, 32, 64, 128, 192):
    abig = torch.randn(R, K + pad, device=dev).to(bf)
    bbig = (torch.randn(N, K + pad, device=dev) * 0.05).to(bf)
    a, b = abig[:, :K], bbig[:, :K]
    out = torch.empty(R, N, device=dev, dtype=torch.float32 if epi == 'F32' else bf)
    ts = timeit([lambda: ops.gemm_nt(a, b, out, epi, m=M, tile=tile)], iters=10 if M * N * K > 2 ** 36 else 20)
    res.append((pad, ts[0]))
    del abig, bbig, a, b, out
  fl = 2.0 * M * N * K
  print('%6d x %5d x %5d tile %2d: ' % (M, N, K, tile) + '  '.join('pad %3d: %7.1f us %5.0f TF' % (p, t, fl / t / 1e6) for p, t in res))
