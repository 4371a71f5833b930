"""Fused Adam over the engine's flat parameter buffer (one HIP launch for the whole video side).

Semantics = torch.optim.Adam (train.py:100 of the reference uses Adam lr 5e-5).  The step counter lives
on the device so the launch is hipGraph-capturable.  Parameters outside the flat buffer (text heads,
text tower) keep using a stock torch optimizer -- see `build_optimizers`.
"""
import torch

from . import _lib, ops
from ._lib import check


class FlatAdam:

  def __init__(self, flat, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
    self.exp_avg = self.exp_avg_sq = self.step_dev = None

  def _grad(self):
    f = self.flat
    g = f.current_grad()
    lo, hi = g.data_ptr(), g.data_ptr() + 4 * f.count
    p0 = next((p for p in f.params if p.requires_grad), None)
    if p0 is not None and p0.grad is not None and not (lo <= p0.grad.data_ptr() < hi):
      # autograd cloned instead of adopting our views (e.g. gradient accumulation): gather them
      for p in f.params:
        if p.grad is not None:
          f.view(p, g).copy_(p.grad)
    return g

  def zero_grad(self, set_to_none=True):
    for p in self.flat.params:
      if set_to_none:
        p.grad = None
      elif p.grad is not None:
        p.grad.zero_()

  @torch.no_grad()
  def step(self):
    f = self.flat
    if self.exp_avg is None or self.exp_avg.device != f.master.device:
      self.exp_avg = torch.zeros_like(f.master)
      self.exp_avg_sq = torch.zeros_like(f.master)
      self.step_dev = torch.zeros(1, dtype=torch.int32, device=f.master.device)
    g = self._grad()
    self.step_dev.add_(1)
    check(_lib.lib().mmt_adam_step(ops._p(f.master), ops._p(g), ops._p(self.exp_avg), ops._p(self.exp_avg_sq),
                                   f.count, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                   ops._p(self.step_dev), ops._stream()), 'mmt_adam_step')
    f._dirty = True  # the bf16 shadows are stale now (the kernel wrote through raw pointers)


def build_optimizers(model, lr=5e-5, **kw):
  """(FlatAdam for the engine parameters, torch Adam for the rest or None)."""
  flat_ids = {id(p) for p in model.engine_params()}
  rest = [p for p in model.parameters() if p.requires_grad and id(p) not in flat_ids]
  return FlatAdam(model._flat, lr=lr, **kw), (torch.optim.Adam(rest, lr=lr, **kw) if rest else None)
