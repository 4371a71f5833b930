"""Global similarity + loss at the batch sizes of 1..8 data-parallel ranks (n = 32..256 pairs): forward, loss, backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import _lib, ops  # noqa: E402
from mmt_amd._lib import check  # noqa: E402
from tools.gemm_lab import timeit  # noqa: E402

L = _lib.lib()
m, d = 7, 512
for n in (32, 64, 128, 256, 512):
  dev = 'cuda'
  txt = torch.nn.functional.normalize(torch.randn(n, m, d, device=dev), dim=-1)
  vid = torch.nn.functional.normalize(torch.randn(n, m, d, device=dev), dim=-1)
  tw = torch.softmax(torch.randn(n, m, device=dev), -1)
  vw = torch.full((n, m), 1.0 / m, device=dev)
  sims, dots = torch.empty(n, n, device=dev), torch.empty(n, n, m, device=dev)
  loss, grad, scratch = torch.empty((), device=dev), torch.empty(n, n, device=dev), torch.empty(3 * n, device=dev)
  dtxt, dvid, dtw, dvw = (torch.empty_like(x) for x in (txt, vid, tw, vw))
  p = ops._p

  def fwd():
    check(L.mmt_sims_fwd(p(txt), p(vid), p(tw), p(vw), n, n, m, d, p(sims), p(dots), ops._stream()), 'f')

  def lossf():
    check(L.mmt_maxmargin(p(sims), n, 0.05, 1, p(scratch), p(loss), p(grad), ops._stream()), 'l')

  def bwd():
    check(L.mmt_sims_bwd(p(txt), p(vid), p(tw), p(vw), p(dots), p(grad), n, n, m, d, p(dtxt), p(dvid), p(dtw), p(dvw),
                         ops._stream()), 'b')

  torch.cuda.synchronize()
  t = timeit([fwd, lossf, bwd])
  print('n %4d | sims_fwd %7.1f us  maxmargin %6.1f us  sims_bwd %7.1f us' % ((n,) + tuple(t)))
