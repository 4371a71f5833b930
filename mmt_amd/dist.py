"""Data parallelism over the batch dimension: one process per GPU, RCCL (torch.distributed 'nccl').

The reference's only multi-GPU mode is nn.DataParallel with `out='embds'`: replicas return embeddings
and the trainer builds the GLOBAL-batch similarity and loss (trainer/trainer.py:134,185-199).  The
MI355X equivalent has one exchange step and one reduction:

  1. all-gather of the per-rank embeddings / weights (a few hundred KB) -> every rank forms the global
     (n x n) similarity and the loss (redundantly: n <= a few hundred pairs, microseconds);
     backward of the gather hands each rank the gradient slice of its own rows -- no reduce-scatter is
     needed because every rank holds the complete loss;
  2. all-reduce(SUM) over the engine's flat gradient buffer (17.1 M fp32 for config B) -- as ONE collective after the
     backward (`GradSync`), or, in `train_step.GraphedTrainStep`'s staged mode, span by span while the rest of the
     backward still runs (the flat layout is ordered back to front for that, `CENet.grad_regions`).  SUM, not mean: the
     loss is already a mean over the global batch, exactly as DataParallel's reduce-add.  Parameters outside the flat
     buffers (a foreign text tower) travel in one extra bucket; a native text tower brings its own flat buffer.

BatchNorm in the text heads uses per-rank statistics, as the reference's replicas do.
Works with the gloo backend on CPU tensors too (used by the world_size-2 tests).
"""
import torch
import torch.distributed as dist


class _AllGatherRows(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, group):
    world = dist.get_world_size(group)
    ctx.rank, ctx.rows = dist.get_rank(group), x.shape[0]
    outs = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(outs, x.contiguous(), group=group)
    return torch.cat(outs, 0)

  @staticmethod
  def backward(ctx, g):
    return g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows].contiguous(), None


def all_gather_rows(x, group=None):
  """Concatenate `x` over ranks along dim 0 (differentiable; every rank must then compute the same loss)."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return x
  return _AllGatherRows.apply(x, group)


def gather_embeddings(embds, group=None):
  """embds: the dict CENet returns for out='embds' -> the same dict for the global batch."""
  return {k: all_gather_rows(v, group) for k, v in embds.items()}


class GradSync:
  """All-reduce(SUM) of gradients: the flat engine buffer as one bucket, remaining params as another."""

  def __init__(self, flat=None, other_params=(), group=None):
    self.flat, self.other, self.group = flat, [p for p in other_params if p.requires_grad], group
    self._bucket = None

  def sync(self, force=False):
    if not dist.is_initialized() or (dist.get_world_size(self.group) == 1 and not force):
      return
    handles = []
    if self.flat is not None:
      g = self.flat.current_grad()
      handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
    grads = [p.grad for p in self.other if p.grad is not None]
    if grads:
      n = sum(g.numel() for g in grads)
      if self._bucket is None or self._bucket.numel() != n or self._bucket.device != grads[0].device:
        self._bucket = torch.empty(n, device=grads[0].device, dtype=grads[0].dtype)
      torch.cat([g.reshape(-1) for g in grads], out=self._bucket)
      handles.append(dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
    for h in handles:
      h.wait()
    if grads:
      o = 0
      for g in grads:
        g.copy_(self._bucket[o:o + g.numel()].view_as(g))
        o += g.numel()


def broadcast_parameters(module, src=0, group=None):
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return
  for t in list(module.parameters()) + list(module.buffers()):
    dist.broadcast(t.data, src=src, group=group)
