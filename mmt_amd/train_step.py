"""One data-parallel training step of the hot path, captured as HIP graphs.

Eagerly, a step issues ~600 tiny launches (text heads, autograd glue, optimizer) and is host-bound at
~7 ms while the kernels need < 2 ms.  `GraphedTrainStep` captures the compute in three HIP graphs and
keeps the two RCCL collectives OUTSIDE of them (so nothing depends on collective capture support):

    graph A : CENet forward (out='embds') on the local batch
    eager   : all-gather of embeddings / weights over ranks            (skipped at world size 1)
    graph B : global similarity + loss + backward (through the gather, into the local embeddings,
              through graph A's autograd graph into the flat gradient buffer)
    eager   : all-reduce(SUM) of the flat gradient buffer + text-head bucket   (skipped at world size 1)
    graph C : optimizer step (fused flat Adam + capturable torch Adam for the rest)

Everything data-dependent lives in device memory (live row count of the token packing, dropout seed,
Adam step counter), so replays are correct for new minibatches copied into the static input buffers.
"""
import torch
import torch.distributed as dist

from . import dist as mdist
from .model import cross_view_similarity
from .optim import FlatAdam


class FlatMinibatch(dict):
  """A minibatch (dict of tensors / dicts of tensors) laid out in ONE device buffer, so that loading it into the
  static input buffers of the captured graphs is a single device-to-device (or host-to-device) copy instead of
  ~45 tiny ones (the reference's move_dict_to_device, trainer/trainer.py:36-52, moves tensor by tensor)."""

  def __init__(self, minibatch, device):
    super().__init__()
    leaves = []

    def walk(d, out):
      for k, v in d.items():
        if isinstance(v, dict):
          out[k] = {}
          walk(v, out[k])
        elif torch.is_tensor(v):
          leaves.append((out, k, v))
        else:
          out[k] = v

    walk(minibatch, self)
    off = 0
    spans = []
    for _, _, v in leaves:
      off = (off + 255) // 256 * 256
      spans.append(off)
      off += v.numel() * v.element_size()
    self.flat = torch.zeros(max(off, 1), dtype=torch.uint8, device=device)
    for (out, k, v), o in zip(leaves, spans):
      n = v.numel() * v.element_size()
      view = self.flat[o:o + n].view(v.dtype).view(v.shape)
      view.copy_(v)
      out[k] = view


def _copy_tree(dst, src):
  for k, v in src.items():
    if isinstance(v, dict):
      _copy_tree(dst[k], v)
    elif torch.is_tensor(v):
      dst[k].copy_(v, non_blocking=True)


class GraphedTrainStep:

  def __init__(self, model, loss_fn, minibatch, lr=5e-5, group=None, use_graphs=True, warmup_steps=3):
    """minibatch: dict of DEVICE tensors as CENet.forward takes them (used as the static input buffers)."""
    self.model, self.loss_fn, self.group = model, loss_fn, group
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.rank = dist.get_rank(group) if dist.is_initialized() else 0
    self.static = minibatch
    flat_ids = {id(p) for p in model.engine_params()}
    rest = [p for p in model.parameters() if p.requires_grad and id(p) not in flat_ids]
    self.opt_flat = FlatAdam(model._flat, lr=lr)
    self.opt_rest = torch.optim.Adam(rest, lr=lr, capturable=use_graphs) if rest else None
    self.sync = mdist.GradSync(model._flat, rest, group)
    self.use_graphs = use_graphs
    self.loss = None
    self._graphs = None
    # Warm-up AND capture run on one dedicated side stream: autograd's AccumulateGrad nodes remember the
    # stream they were created on, and a node bound to the default stream breaks capture of the backward.
    self._stream = torch.cuda.Stream()
    self._stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(self._stream):
      for _ in range(warmup_steps):  # allocates every lazily created buffer / optimizer state
        self._eager_step()
    torch.cuda.current_stream().wait_stream(self._stream)
    if use_graphs:
      self._capture()

  # ---- pieces ------------------------------------------------------------------------------------
  def _forward(self):
    mb = self.static
    return self.model(mb['token_ids'], mb['features'], mb['features_t'], mb['features_ind'], mb['features_avgpool'],
                      mb['features_maxpool'], mb['query_masks'], out='embds')

  def _gather(self, e):
    """-> dict of LEAF tensors holding the global batch (requires_grad where the local one does)."""
    out = {}
    for k, v in e.items():
      if self.world == 1:
        g = v.detach()
      else:
        g = self._gbuf[k] if self._gbuf is not None else torch.empty((self.world * v.shape[0],) + v.shape[1:],
                                                                      device=v.device, dtype=v.dtype)
        dist.all_gather_into_tensor(g, v.detach().contiguous(), group=self.group)
      out[k] = g
    return out

  def _loss_backward(self, e, g):
    fast = self._fast_loss_backward(e, g)
    if fast is not None:
      return fast
    leaves = {k: v.detach().requires_grad_(e[k].requires_grad) for k, v in g.items()}
    sims = cross_view_similarity(leaves['vid_embds'], leaves['text_embds'], leaves['vid_weights'],
                                 leaves['text_weights'], 'avg')
    loss = self.loss_fn(sims)
    need = [k for k in leaves if leaves[k].requires_grad]
    grads = torch.autograd.grad(loss, [leaves[k] for k in need])
    b = e['vid_embds'].shape[0]
    sl = slice(self.rank * b, (self.rank + 1) * b)
    torch.autograd.backward([e[k] for k in need], [gr[sl] for gr in grads])
    return loss.detach()

  def _fast_loss_backward(self, e, g):
    """Our own similarity + loss kernels called directly (no autograd bookkeeping for this tiny sub-graph: saves the
    ones_like / multiply / slice launches): global sims -> loss + dL/dsims in one kernel -> similarity backward ->
    the local rows' gradients pushed into graph A's autograd graph.  None = not applicable (foreign loss module,
    several captions per video): the generic autograd path is used."""
    import ctypes

    from . import _lib, ops
    from .loss import InfoNceLoss, MaxMarginRankingLoss
    from ._lib import check
    if not isinstance(self.loss_fn, (MaxMarginRankingLoss, InfoNceLoss)) or g['text_embds'].shape[2] != 1:
      return None
    if g['vid_weights'].requires_grad or e['vid_weights'].requires_grad:
      return None
    L = _lib.lib()
    vid = g['vid_embds'].detach().contiguous().float()
    n, m, d = vid.shape
    txt = g['text_embds'].detach().reshape(n, m, d).contiguous().float()  # C == 1: (n, M, 1, d) -> (n, M, d) is a view
    tw = g['text_weights'].detach().reshape(n, m).contiguous().float()
    vw = g['vid_weights'].detach().reshape(n, m).contiguous().float()
    dev = vid.device
    sims = torch.empty(n, n, device=dev, dtype=torch.float32)
    dots = torch.empty(n, n, m, device=dev, dtype=torch.float32)
    check(L.mmt_sims_fwd(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), n, n, m, d, ops._p(sims), ops._p(dots),
                         ops._stream()), 'mmt_sims_fwd')
    loss = torch.empty((), device=dev, dtype=torch.float32)
    grad = torch.empty(n, n, device=dev, dtype=torch.float32)
    scratch = torch.empty(3 * n, device=dev, dtype=torch.float32)
    if isinstance(self.loss_fn, MaxMarginRankingLoss):
      check(L.mmt_maxmargin(ops._p(sims), n, float(self.loss_fn.margin), int(self.loss_fn.fix_norm), ops._p(scratch),
                            ops._p(loss), ops._p(grad), ops._stream()), 'mmt_maxmargin')
    else:
      check(L.mmt_infonce(ops._p(sims), n, ops._p(scratch), ops._p(loss), ops._p(grad), ops._stream()), 'mmt_infonce')
    dtxt, dvid, dtw, dvw = (torch.empty_like(x) for x in (txt, vid, tw, vw))
    check(L.mmt_sims_bwd(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), ops._p(dots), ops._p(grad), n, n, m, d,
                         ops._p(dtxt), ops._p(dvid), ops._p(dtw), ops._p(dvw), ops._stream()), 'mmt_sims_bwd')
    b = e['vid_embds'].shape[0]
    sl = slice(self.rank * b, (self.rank + 1) * b)
    outs, grads = [], []
    for k, gfull in (('vid_embds', dvid), ('text_embds', dtxt.view(n, m, 1, d)), ('text_weights', dtw.view(n, 1, m))):
      if e[k].requires_grad:
        outs.append(e[k])
        grads.append(gfull[sl])
    torch.autograd.backward(outs, grads)
    return loss

  def _zero(self):
    self.opt_flat.zero_grad()
    if self.opt_rest is not None:
      self.opt_rest.zero_grad(set_to_none=True)

  def _opt(self):
    self.opt_flat.step()
    if self.opt_rest is not None:
      self.opt_rest.step()

  _gbuf = None

  def _eager_step(self):
    self._zero()
    e = self._forward()
    g = self._gather(e)
    self.loss = self._loss_backward(e, g)
    self.sync.sync()
    self._opt()

  # ---- capture -----------------------------------------------------------------------------------
  def _capture(self):
    torch.cuda.synchronize()
    self._zero()
    ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga, stream=self._stream):
      e = self._forward()
    pool = ga.pool()
    if self.world > 1:
      self._gbuf = {k: torch.empty((self.world * v.shape[0],) + v.shape[1:], device=v.device, dtype=v.dtype)
                    for k, v in e.items()}
    with torch.cuda.stream(self._stream):
      g = self._gather(e)
    with torch.cuda.graph(gb, pool=pool, stream=self._stream):
      self.loss = self._loss_backward(e, g)
    with torch.cuda.graph(gc, pool=pool, stream=self._stream):
      self._opt()
    self._graphs, self._e = (ga, gb, gc), e
    torch.cuda.synchronize()

  # ---- public ------------------------------------------------------------------------------------
  def load(self, minibatch):
    """Copy a new minibatch (device tensors, same shapes) into the static input buffers."""
    if isinstance(minibatch, FlatMinibatch) and isinstance(self.static, FlatMinibatch) \
        and minibatch.flat.numel() == self.static.flat.numel():
      self.static.flat.copy_(minibatch.flat, non_blocking=True)
    else:
      _copy_tree(self.static, minibatch)

  def step(self):
    """Runs one optimisation step on the current static inputs; returns the (device) loss tensor."""
    if not self.use_graphs:
      self._eager_step()
      return self.loss
    ga, gb, gc = self._graphs
    ga.replay()
    if self.world > 1:
      self._gather(self._e)
    gb.replay()
    self.sync.sync()
    gc.replay()
    return self.loss
