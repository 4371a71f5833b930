"""Drop-ins for the reference's `model/loss.py` on the MI355X kernels.

`MaxMarginRankingLoss(margin, fix_norm)` and `InfoNceLoss()` keep the reference constructor and
`forward(x)` (x = (n, n) similarity matrix, rows = text, cols = video; model/loss.py:29-81) and return a
0-dim tensor supporting `.backward()` / `.item()`.  The reference builds 2n^2-long index vectors on the
host every step (loss.py:55-63); here one kernel produces the loss and its gradient matrix.
"""
import torch
from torch import nn

from . import _lib, ops
from ._lib import check


class _LossFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, kind, margin, fix_norm):
    if not x.is_cuda:
      raise RuntimeError('mmt_amd losses run on the GPU only (no CPU fallback)')
    n = x.shape[0]
    if x.dim() != 2 or x.shape[1] != n:
      raise ValueError('expected a square similarity matrix')
    xs = x.detach().contiguous().float()
    loss = torch.empty((), device=x.device, dtype=torch.float32)
    grad = torch.empty(n, n, device=x.device, dtype=torch.float32)
    scratch = torch.empty(3 * n, device=x.device, dtype=torch.float32)
    L = _lib.lib()
    if kind == 0:
      check(L.mmt_maxmargin(ops._p(xs), n, float(margin), int(fix_norm), ops._p(scratch), ops._p(loss), ops._p(grad),
                            ops._stream()), 'mmt_maxmargin')
    else:
      check(L.mmt_infonce(ops._p(xs), n, ops._p(scratch), ops._p(loss), ops._p(grad), ops._stream()), 'mmt_infonce')
    ctx.save_for_backward(grad)
    ctx.in_dtype = x.dtype
    return loss

  @staticmethod
  def backward(ctx, gout):
    grad, = ctx.saved_tensors
    return (grad * gout).to(ctx.in_dtype), None, None, None


class MaxMarginRankingLoss(nn.Module):
  """model/loss.py:29-65."""

  def __init__(self, margin=1, fix_norm=True):
    super().__init__()
    self.fix_norm = fix_norm
    self.margin = margin

  def forward(self, x):
    return _LossFn.apply(x, 0, self.margin, self.fix_norm)


class InfoNceLoss(nn.Module):
  """model/loss.py:68-81."""

  def forward(self, x):
    return _LossFn.apply(x, 1, 0.0, True)
