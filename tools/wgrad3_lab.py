"""wgrad3.hip on configs[4]'s shapes (four weight gradients of a d = 1024 layer, 14 393 live rows): time per launch.  Lab
builds (MMT_LAB_DEFINES=MMT_W3_LAB_NOREADS | MMT_W3_LAB_PLAINREADS | MMT_W3_LAB_NOMFMA python -m mmt_amd.build --instr) switch
parts of the loop off; results are then wrong, only the time is read.   MMT_HIP_LIB=... python tools/wgrad3_lab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402
from tools.gemm_lab import timeit  # noqa: E402

dev = torch.device('cuda:0')
rows, live, d, inter = 14464, 14393, 1024, 6144
items = []
for (N, K2) in [(inter, d), (d, inter), (3 * d, d), (d, d)]:
  a = (torch.randn(rows, N, device=dev) * 0.5).to(torch.bfloat16)
  b = (torch.randn(rows, K2, device=dev) * 0.5).to(torch.bfloat16)
  items.append((a, b, torch.empty(N, K2, device=dev), torch.empty(N, device=dev)))
nrd = torch.tensor([live], device=dev, dtype=torch.int32)
ts = timeit([lambda: ops.wgrad_grouped(items, rows, n_rows_dev=nrd)], iters=5)
fl = 2.0 * live * (2 * inter * d + 4 * d * d)
print('%s: %.1f us per launch = %.0f TFLOP/s = %.3f of 2.5 PF' % (os.environ.get('MMT_HIP_LIB', 'default lib'), ts[0], fl / ts[0] / 1e6, fl / ts[0] / 1e6 / 2500))
