"""Per-rank work of BASELINE.json configs[4] (HowTo100M-scale: 64k-pair similarity + max-margin over 8 ranks):
one rank's row block -- b = 8192 texts x n = 65536 videos, M = 7 experts, d = 1024 -- forward + backward on one MI355X,
with the cross-rank quantities (global diagonal, column counts) stood in by their single-block values.
   python tools/large_sim_bench.py [--b 8192 --n 65536 --d 1024]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd.large_sim import RowBlock  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--b', type=int, default=8192)
ap.add_argument('--n', type=int, default=65536)
ap.add_argument('--m', type=int, default=7)
ap.add_argument('--d', type=int, default=1024)
ap.add_argument('--iters', type=int, default=3)
a = ap.parse_args()
dev = torch.device('cuda:0')
g = torch.Generator(device='cuda').manual_seed(0)
nrm = lambda x: torch.nn.functional.normalize(x, dim=-1)
vid = nrm(torch.randn(a.n, a.m, a.d, device=dev, generator=g))
txt = nrm(torch.randn(a.b, a.m, a.d, device=dev, generator=g) + 0.3 * vid[:a.b])
tw = torch.softmax(torch.randn(a.b, a.m, device=dev, generator=g), -1)
vw = torch.full((a.n, a.m), 1.0 / a.m, device=dev)
torch.cuda.synchronize()
best = None
for it in range(a.iters):
  t0 = time.perf_counter()
  blk = RowBlock(txt, tw, vid, vw, 0, 0.05)
  diag_l = blk.phase_similarity()
  diag = torch.zeros(a.n, device=dev)
  diag[:a.b] = diag_l
  colcnt, loss = blk.phase_counts(diag)
  dtxt, dtw, q = blk.phase_backward(colcnt)
  dvid = blk.phase_video_grad(q[:a.b], vid[:a.b], vw[:a.b])
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  best = dt if best is None else min(best, dt)
flops = 3 * 2.0 * a.b * a.n * a.m * a.d
assert torch.isfinite(loss) and torch.isfinite(dtxt).all() and torch.isfinite(dvid).all()
# size-independent properties: the gradient wrt S sums to zero row-block-wise up to the column hinges owned elsewhere;
# unit-norm inputs => |S| <= 1; tw sums to one => dtw is orthogonal to the all-ones direction up to the normaliser term
smax = blk.similarity().abs().max().item()
assert smax <= 1.0 + 2e-3
print('row block %d x %d, M=%d, d=%d: %.1f ms fwd+bwd (%.0f TFLOP/s on the three GEMMs), loss %.5f, max|S| %.4f, peak mem %.1f GB'
      % (a.b, a.n, a.m, a.d, best * 1e3, flops / best / 1e12, loss.item(), smax,
         torch.cuda.max_memory_allocated() / 2 ** 30))
