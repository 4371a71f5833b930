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


def gather_stray_grads(flat):
  """`FlatParams.current_grad()` holds the step's gradients only where autograd ADOPTED the engine's views as `.grad`.
  Under gradient accumulation (or when AccumulateGrad clones) `.grad` lives elsewhere: copy it into the flat buffer --
  BEFORE any reduction, so that the optimizer later finds reduced values there and does not overwrite them with the
  unreduced local ones."""
  g = flat.current_grad()
  lo, hi = g.data_ptr(), g.data_ptr() + 4 * flat.count
  for p in flat.params:
    if p.grad is not None and not (lo <= p.grad.data_ptr() < hi):
      flat.view(p, g).copy_(p.grad)
      p.grad = flat.view(p, g)  # from here on the flat buffer IS the gradient
  return g


class _Both:
  """Two collective handles waited in issue order."""

  def __init__(self, *works):
    self.works = [w for w in works if w is not None]

  def wait(self):
    for w in self.works:
      w.wait()


class _Then:
  """`first`, then (once it has completed) a second collective issued by `issue()`; waits for both."""

  def __init__(self, first, issue):
    self.first, self.issue = first, issue

  def wait(self):
    if self.first is not None:
      self.first.wait()
    w = self.issue()
    if w is not None:
      w.wait()


class WireBuffer:
  """How a span of the flat gradient buffer crosses the wire.

  dtype: torch.bfloat16 = bf16 wire format: the all-reduce moves half the bytes (68 MB -> 34 MB for the video side; ring
  time over one xGMI link per hop halves with it).  The sum is formed in bf16 by the collective; the fp32 gradient
  buffer is overwritten with the result.  One rounding of every addend + (world - 1) bf16 additions.

  algo: 'allreduce' = one all_reduce(SUM) per span (RCCL picks ring / tree); 'rs_ag' = reduce_scatter_tensor of the span
  (zero-padded to a multiple of the world size) followed by all_gather_into_tensor of the reduced shards -- the
  decomposition a direct full-mesh exchange over all 7 xGMI links maps to (~0.11 ms for 68 MB
  against ~0.8 ms for a ring bound by one link per hop).  Same sums up to fp32 addition order (bit-identical on 2 ranks)."""

  def __init__(self, dtype=None, algo='allreduce'):
    if algo not in ('allreduce', 'rs_ag'):
      raise ValueError("WireBuffer algo: 'allreduce' or 'rs_ag'")
    self.dtype, self.algo = dtype, algo
    self._bufs = {}

  def _buf(self, span, numel, dtype):
    key = (span.data_ptr(), span.numel(), numel, dtype)
    buf = self._bufs.get(key)
    if buf is None:
      buf = self._bufs[key] = torch.zeros(numel, device=span.device, dtype=dtype)
    return buf

  def _shard_geometry(self, n, world):
    per = -(-n // (4 * world)) * 4  # elements per rank, a multiple of 4 (16-byte aligned shard starts)
    return per

  def reduce_scatter(self, span, group, async_op=True):
    """First half of 'rs_ag' on its own: -> (work, shard, send buffer, lo, cnt).  Once `work` has completed, `shard` (wire
    dtype, `per` elements, persistent per span) holds the SUM over ranks of span[lo : lo + per], lo = rank * per; its first
    cnt elements lie inside the span (the tail of the last rank's shard is padding).  `shard_f32` widens a bf16 shard."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    wire = self.dtype if self.dtype not in (None, torch.float32) else None
    n = span.numel()
    per = self._shard_geometry(n, world)
    dt = wire or torch.float32
    if wire is None and n == per * world:
      buf = span
    else:
      buf = self._buf(span, per * world, dt)  # (the tail beyond n is zero on every rank: the sums there stay zero)
      buf[:n].copy_(span)
    key = ('shard', span.data_ptr(), per, dt)
    shard = self._bufs.get(key)  # this rank's reduced shard: its own buffer (no send / receive aliasing)
    if shard is None:
      shard = self._bufs[key] = torch.empty(per, device=span.device, dtype=dt)
    work = dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    lo = rank * per
    return work, shard, buf, lo, max(0, min(per, n - lo))

  def shard_f32(self, span, shard):
    """The reduced shard as fp32 (a persistent widened copy when the wire is bf16); call after the collective completed."""
    if shard.dtype == torch.float32:
      return shard
    key = ('shard32', span.data_ptr(), shard.numel())
    out = self._bufs.get(key)
    if out is None:
      out = self._bufs[key] = torch.empty(shard.numel(), device=shard.device, dtype=torch.float32)
    out.copy_(shard)
    return out

  def all_gather_span(self, span, group, async_op=True, scratch=False):
    """span (fp32 view of a flat buffer) holds valid values in this rank's shard [rank * per, ...) only: all-gather the
    shards so that every rank holds the whole span.  -> (work, finish).  scratch=True: a one-off exchange (an optimizer
    checkpoint's moment gather) -- the padded gather / send buffers are temporaries that die with `finish` instead of
    being cached per span for the life of the run (they would add ~2x the Adam-moment footprint to device memory)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = span.numel()
    per = self._shard_geometry(n, world)
    if n == per * world:
      mine = span[rank * per:(rank + 1) * per]
      if dist.get_backend(group) == 'nccl':
        return dist.all_gather_into_tensor(span, mine, group=group, async_op=async_op), (lambda: None)
      mine = mine.clone()  # (gloo: no in-place form)
      return dist.all_gather_into_tensor(span, mine, group=group, async_op=async_op), (lambda: None)
    if scratch:
      buf = torch.empty(per * world, device=span.device, dtype=torch.float32)
      mine = torch.zeros(per, device=span.device, dtype=torch.float32)
    else:
      buf = self._buf(span, per * world, torch.float32)
      key = ('wshard', span.data_ptr(), per)
      mine = self._bufs.get(key)
      if mine is None:
        mine = self._bufs[key] = torch.zeros(per, device=span.device, dtype=torch.float32)
    lo = rank * per
    cnt = max(0, min(per, n - lo))
    if cnt:
      mine[:cnt].copy_(span[lo:lo + cnt])
    work = dist.all_gather_into_tensor(buf, mine, group=group, async_op=async_op)
    return work, (lambda: span.copy_(buf[:n]))

  def reduce(self, span, group, async_op=True):
    """span: fp32 view of the flat gradient buffer.  -> (work handle, finish callable)"""
    wire = self.dtype if self.dtype not in (None, torch.float32) else None
    world = dist.get_world_size(group)
    if self.algo == 'rs_ag' and world > 1:
      n = span.numel()
      rs, shard, buf, _, _ = self.reduce_scatter(span, group, async_op)
      fin = (lambda: None) if buf is span else (lambda: span.copy_(buf[:n]))
      if dist.get_backend(group) == 'nccl':
        # both collectives run in issue order on the communicator's stream: no host round trip between them
        ag = dist.all_gather_into_tensor(buf, shard, group=group, async_op=async_op)
        return _Both(rs, ag), fin
      return _Then(rs, lambda: dist.all_gather_into_tensor(buf, shard, group=group, async_op=async_op)), fin
    if wire is None:
      h = dist.all_reduce(span, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
      return h, (lambda: None)
    buf = self._buf(span, span.numel(), wire)
    buf.copy_(span)
    h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return h, (lambda: span.copy_(buf))


class GradSync:
  """All-reduce(SUM) of gradients: the flat engine buffer as one bucket, remaining params as another.
  grad_dtype=torch.bfloat16 sends the flat buffer as bf16 (`WireBuffer`)."""

  def __init__(self, flat=None, other_params=(), group=None, grad_dtype=None, algo='allreduce'):
    self.flat, self.other, self.group = flat, [p for p in other_params if p.requires_grad], group
    self._bucket = None
    self.wire = WireBuffer(grad_dtype, algo)

  def sync(self, force=False):
    if not dist.is_initialized() or (dist.get_world_size(self.group) == 1 and not force):
      return
    handles, finish = [], []
    if self.flat is not None:
      g = gather_stray_grads(self.flat)
      h, fin = self.wire.reduce(g, self.group)
      handles.append(h)
      finish.append(fin)
    grads = [p.grad for p in self.other if p.grad is not None]
    if grads:
      n = sum(g.numel() for g in grads)
      if self._bucket is None or self._bucket.numel() != n or self._bucket.device != grads[0].device:
        self._bucket = torch.empty(n, device=grads[0].device, dtype=grads[0].dtype)
      torch.cat([g.reshape(-1) for g in grads], out=self._bucket)
      handles.append(dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
    for h in handles:
      if h is not None:
        h.wait()
    for fin in finish:
      fin()
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
