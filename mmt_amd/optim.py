"""Fused Adam over the engine's flat parameter buffer (one HIP launch for the whole video side).

Semantics = torch.optim.Adam (train.py:100 of the reference uses Adam lr 5e-5).  The step counter lives
on the device so the launch is hipGraph-capturable.  Parameters outside the flat buffer (text heads,
text tower) keep using a stock torch optimizer -- see `build_optimizers`.
"""
import torch

from . import _lib, ops
from ._lib import check


def ctypes_ptr(t, offset):
  """Raw pointer to element `offset` of the fp32 tensor `t`."""
  import ctypes
  return ctypes.c_void_p(t.data_ptr() + 4 * int(offset))


class FlatAdam:
  """`param_groups` is shaped like a torch optimizer's, so scheduler code that only touches
  `optimizer.param_groups[i]['lr']` (the reference's StepLR + LinearWarmup, train.py:101-103,
  trainer/trainer.py:150-160) drives it.  The rate is mirrored into a device scalar that the kernel reads, so a captured
  optimizer graph follows the schedule: call `sync_lr()` (cheap, no-op when unchanged) before replaying --
  `GraphedTrainStep.step` does.  `state_dict()` / `load_state_dict()` use torch.optim.Adam's layout (per-parameter
  `step` / `exp_avg` / `exp_avg_sq`, parameters numbered in `param_groups` order), so the reference's checkpoints
  (base/base_trainer.py:353-365 saves `optimizer.state_dict()`, :426-432 restores it) round-trip, also to and from a
  stock torch.optim.Adam over the same parameter list.  A checkpoint of the REFERENCE holds ONE Adam over every trainable
  model parameter (train.py:95-100); `merged_state_dict` / `load_merged_state_dict` translate between that layout and
  the several optimizers of a step (one FlatAdam per flat buffer + a torch Adam for the rest)."""

  def __init__(self, flat, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    self.flat, self.betas, self.eps, self.weight_decay = flat, betas, eps, weight_decay
    self.exp_avg = self.exp_avg_sq = self.step_dev = None
    self.param_groups = [dict(params=list(flat.params), lr=lr, initial_lr=lr, betas=betas, eps=eps,
                              weight_decay=weight_decay)]
    self.lr_dev, self._lr_on_dev = None, None

  @property
  def lr(self):
    return self.param_groups[0]['lr']

  @lr.setter
  def lr(self, value):
    self.param_groups[0]['lr'] = value

  def sync_lr(self):
    """Push param_groups[0]['lr'] to the device scalar if it changed (stream-ordered fill, outside any graph)."""
    lr = float(self.lr)
    if self.lr_dev is not None and lr != self._lr_on_dev and not (self.lr_dev.is_cuda and torch.cuda.is_current_stream_capturing()):
      self.lr_dev.fill_(lr)
      self._lr_on_dev = lr

  def _frozen_spans(self):
    """(offset, count) runs of the flat layout whose parameters have requires_grad == False.  The engine writes a
    gradient for every parameter it owns; a frozen one (txt_agg='bertfrz...', model/model.py:164-192) must not move,
    and with zero gradient from step one Adam's update is exactly zero (weight_decay must be 0)."""
    f = self.flat
    key = tuple(p.requires_grad for p in f.params)
    if getattr(self, '_frozen_key', None) != key:
      spans = []
      for p in f.params:
        if not p.requires_grad:
          o, n = f.offset(p), p.numel()
          if spans and spans[-1][0] + spans[-1][1] >= o - 64:
            spans[-1] = (spans[-1][0], o + n - spans[-1][0])
          else:
            spans.append((o, n))
      if spans and self.weight_decay != 0.0:
        raise NotImplementedError('frozen parameters with weight_decay != 0')
      self._frozen_key, self._frozen = key, spans
    return self._frozen

  def _grad(self):
    f = self.flat
    g = f.current_grad()
    for o, n in self._frozen_spans():
      g[o:o + n].zero_()
    # autograd cloned instead of adopting our views (e.g. gradient accumulation): gather them.  A data-parallel
    # reduction (GradSync) has done this BEFORE reducing and re-pointed .grad at the flat buffer, so reduced values
    # are never overwritten here.
    from .dist import gather_stray_grads
    return gather_stray_grads(f)

  def _ensure_state(self):
    f = self.flat
    if f.master is None:
      raise RuntimeError('FlatAdam: the parameters are not on a device yet (FlatParams.ensure)')
    if self.exp_avg is None or self.exp_avg.device != f.master.device:
      self.exp_avg = torch.zeros_like(f.master)
      self.exp_avg_sq = torch.zeros_like(f.master)
      # [steps taken, ticket]: the fused kernel increments the count itself (mmt_adam_step_fused, bump_step)
      self._step_store = torch.zeros(2, dtype=torch.int32, device=f.master.device)
      self.step_dev = self._step_store[:1]
    if self.lr_dev is None or self.lr_dev.device != f.master.device:
      self.lr_dev = torch.full((1,), float(self.lr), dtype=torch.float32, device=f.master.device)
      self._lr_on_dev = float(self.lr)

  def state_dict(self):
    """torch.optim.Adam layout: {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]}."""
    f = self.flat
    state = {}
    if self.exp_avg is not None:
      step = self.step_dev.detach().to(torch.float32).reshape(()).cpu()
      for i, p in enumerate(f.params):
        state[i] = dict(step=step.clone(), exp_avg=f.view(p, self.exp_avg).detach().clone(),
                        exp_avg_sq=f.view(p, self.exp_avg_sq).detach().clone())
    group = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
    group['params'] = list(range(len(f.params)))
    return {'state': state, 'param_groups': [group]}

  def load_state_dict(self, sd):
    f = self.flat
    groups = sd['param_groups']
    if len(groups) != 1 or len(groups[0]['params']) != len(f.params):
      raise ValueError('FlatAdam.load_state_dict: expected one param group of %d parameters' % len(f.params))
    for k, v in groups[0].items():
      if k != 'params':
        self.param_groups[0][k] = v
    self.betas = tuple(self.param_groups[0].get('betas', self.betas))
    self.eps = self.param_groups[0].get('eps', self.eps)
    self.weight_decay = self.param_groups[0].get('weight_decay', self.weight_decay)
    state = sd.get('state', {})
    if state:
      self._ensure_state()
      steps = set()
      with torch.no_grad():
        # parameters without an entry in the checkpoint (never stepped there) must not keep moments of THIS run
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, p in enumerate(f.params):
          st = state.get(i, state.get(str(i)))
          if st is None:
            continue
          f.view(p, self.exp_avg).copy_(st['exp_avg'])
          f.view(p, self.exp_avg_sq).copy_(st['exp_avg_sq'])
          steps.add(int(float(st['step'])))
      if len(steps) > 1:
        raise ValueError('FlatAdam keeps ONE step counter for the flat buffer; the checkpoint holds %s' % sorted(steps))
      if steps:
        self.step_dev.fill_(steps.pop())
    if self.lr_dev is not None:
      self._lr_on_dev = None
      self.sync_lr()

  def zero_grad(self, set_to_none=True):
    for p in self.flat.params:
      if set_to_none:
        p.grad = None
      elif p.grad is not None:
        p.grad.zero_()

  @torch.no_grad()
  def step(self):
    f = self.flat
    self._ensure_state()
    self.sync_lr()
    g = self._grad()
    table = self._segment_table() if self.fuse_shadows else None
    if self._queue is not None and self._queue['armed']:
      # the backward's GEMM launches have been draining the queue (riders); whatever is left + the step count
      if table is None or not self._queue_valid():
        raise RuntimeError('FlatAdam: the armed optimizer queue no longer matches the flat buffers (rebuild it)')
      q = self._queue
      import ctypes
      check(_lib.lib().mmt_adam_step_queue(ctypes.byref(q['host']), ops._p(q['dev']), ops._stream()), 'mmt_adam_step_queue')
      f.shadows_fresh()
      return
    if table is not None:
      # ONE launch: Adam over every segment of the flat buffer + the bf16 W / W^T shadows of the GEMM weights + the step
      # counter (the last block to finish stores steps + 1)
      host, dev, n = table
      check(_lib.lib().mmt_adam_step_fused(ops._p(f.master), ops._p(g), ops._p(self.exp_avg), ops._p(self.exp_avg_sq),
                                           host, ops._p(dev), n, float(self.lr), self.betas[0], self.betas[1], self.eps,
                                           self.weight_decay, ops._p(self._step_store), ops._p(self.lr_dev), 1,
                                           ops._stream()), 'mmt_adam_step_fused')
      f.shadows_fresh()
      return
    self.step_dev.add_(1)
    check(_lib.lib().mmt_adam_step(ops._p(f.master), ops._p(g), ops._p(self.exp_avg), ops._p(self.exp_avg_sq),
                                   f.count, float(self.lr), self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                   ops._p(self.step_dev), ops._p(self.lr_dev), ops._stream()), 'mmt_adam_step')
    f._dirty = True  # the bf16 shadows are stale now (the kernel wrote through raw pointers)

  @torch.no_grad()
  def step_shard(self, offset, count, grad, first):
    """Adam over master[offset : offset + count] with the gradients in `grad` (fp32, >= count elements) -- a data-parallel
    rank's share of the step when the gradient exchange is a reduce-scatter (`GraphedTrainStep(shard_optimizer=True)`:
    train.py:97-103's optimizer run on 1/N of the parameters per rank, the updated weights all-gathered afterwards).  All
    shards of a step share one step count: `first` advances it.  The bf16 shadows are stale afterwards (`flat.pack`)."""
    f = self.flat
    self._ensure_state()
    self.sync_lr()
    if first:
      self.step_dev.add_(1)
    if count <= 0:
      return
    if offset % 4 or count % 4:
      raise ValueError('FlatAdam.step_shard: offset and count must be multiples of 4 elements')
    check(_lib.lib().mmt_adam_step(ctypes_ptr(f.master, offset), ops._p(grad), ctypes_ptr(self.exp_avg, offset),
                                   ctypes_ptr(self.exp_avg_sq, offset), count, float(self.lr), self.betas[0], self.betas[1],
                                   self.eps, self.weight_decay, ops._p(self.step_dev), ops._p(self.lr_dev), ops._stream()),
          'mmt_adam_step')
    f._dirty = True

  @torch.no_grad()
  def step_span(self, offset, count, bump):
    """The fused step over ONE contiguous span of the flat buffer (a `CENet.grad_regions` entry) on the current stream:
    a data-dependence-ordered optimizer -- the span's update runs as soon as its gradients are final, on a side stream
    under the rest of the backward (train_step.GraphedTrainStep, fork mode).  Every span of a step reads the same step
    count; `bump` (the LAST span's launch) advances it.  The spans of one step must tile [0, flat.count)."""
    import ctypes

    from ._lib import MmtAdamSeg
    f = self.flat
    self._ensure_state()
    table = self._segment_table() if self.fuse_shadows else None
    if table is None:
      raise RuntimeError('FlatAdam.step_span needs the fused kernel (bf16 shadows allocated: run a forward first)')
    host, dev, n = table
    idx = [i for i in range(n) if offset <= host[i].offset < offset + count]
    if not idx or idx != list(range(idx[0], idx[-1] + 1)) or host[idx[0]].offset != offset or \
        host[idx[-1]].offset + host[idx[-1]].count != offset + count:
      raise ValueError('FlatAdam.step_span: [%d, %d) does not start and end on segment boundaries' % (offset, offset + count))
    size = ctypes.sizeof(MmtAdamSeg)
    sub = ctypes.cast(ctypes.byref(host, idx[0] * size), ctypes.POINTER(MmtAdamSeg))
    check(_lib.lib().mmt_adam_step_fused(ops._p(f.master), ops._p(f.current_grad()), ops._p(self.exp_avg),
                                         ops._p(self.exp_avg_sq), sub, ctypes.c_void_p(dev.data_ptr() + idx[0] * size),
                                         len(idx), float(self.lr), self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                         ops._p(self._step_store), ops._p(self.lr_dev), 1 if bump else 2, ops._stream()),
          'mmt_adam_step_fused')
    if bump:
      f.shadows_fresh()

  # ---- the optimizer as a work queue the backward's GEMM launches drain (include/mmt_hip.h, "Adam riders") -----------
  _queue = None

  def build_queue(self, stage_of, chain=None):
    """The fused step cut into its units of work (one 64x64 tile of a shadowed matrix / 4096 elements of a plain span),
    ordered by the stage of the backward after which a unit's gradients are final.

    stage_of(param) -> int >= 0 (final once stage s of the caller's backward has run) or None (only final when the whole
    backward has run: the unit stays for `step()`).  A unit that overlaps several parameters takes the latest of their
    stages.  chain: another FlatAdam whose queue the rider blocks of THIS queue's launches drain first (all of it must be
    final by then: the native text tower's leftovers under the video side's backward).
    Returns the number of stages n; `queue_limit(k)` = entries in the first k stages.  While a queue is armed, `step()`
    runs mmt_adam_step_queue (the entries no rider took + the step count) instead of the single fused launch: weights,
    moments and bf16 shadows come out bit-identical (tests/test_optim_gpu.py).  Reference: train.py:100,
    trainer/trainer.py:203-204 (`optimizer.step()` after the whole backward)."""
    import ctypes

    import numpy as np

    from ._lib import MmtAdamQueue, RIDER_STAGES, RIDER_STATE_WORDS
    f = self.flat
    self._ensure_state()
    if self._frozen_spans():
      raise NotImplementedError('optimizer queue with frozen parameters in the flat buffer')
    table = self._segment_table()
    if table is None:
      raise RuntimeError('FlatAdam.build_queue needs the fused kernel (bf16 shadows allocated: run a forward first)')
    host, dev, n = table
    L = _lib.lib()
    # stage of every element range that belongs to a parameter
    starts = np.array([f.offset(p) for p in f.params], dtype=np.int64)
    ends = starts + np.array([p.numel() for p in f.params], dtype=np.int64)
    NEVER = 1 << 30
    pstage = np.array([NEVER if stage_of(p) is None else int(stage_of(p)) for p in f.params], dtype=np.int64)
    order = np.argsort(starts)
    starts, ends, pstage = starts[order], ends[order], pstage[order]

    def stage_of_range(lo, hi):
      i0 = int(np.searchsorted(ends, lo, side='right'))
      i1 = int(np.searchsorted(starts, hi, side='left'))
      return int(pstage[i0:i1].max()) if i1 > i0 else -1  # padding only: final from the start

    seg_ids, blks, stages = [], [], []
    for i in range(n):
      sg = host[i]
      nb = L.mmt_adam_fused_blocks(ctypes.byref(sg))
      if sg.dst:  # a matrix: every tile belongs to the parameters of one shadow (same stage)
        st = stage_of_range(sg.offset, sg.offset + sg.count)
        seg_ids += [i] * nb
        blks += list(range(nb))
        stages += [st] * nb
      else:
        for b in range(nb):
          lo = sg.offset + 4096 * b
          seg_ids.append(i)
          blks.append(b)
          stages.append(stage_of_range(lo, min(lo + 4096, sg.offset + sg.count)))
    stages = np.maximum(np.array(stages, dtype=np.int64), 0)
    perm = np.argsort(stages, kind='stable')
    stages = stages[perm]
    n_stages = int(stages[stages < NEVER].max()) + 1 if (stages < NEVER).any() else 0
    if n_stages > RIDER_STAGES:
      raise ValueError('optimizer queue: %d stages (at most %d)' % (n_stages, RIDER_STAGES))
    devc = f.master.device
    unit_seg = torch.from_numpy(np.array(seg_ids, dtype=np.int32)[perm].copy()).to(devc)
    unit_blk = torch.from_numpy(np.array(blks, dtype=np.int32)[perm].copy()).to(devc)
    state = torch.zeros(RIDER_STATE_WORDS, dtype=torch.int32, device=devc)
    q = MmtAdamQueue()
    q.p, q.m, q.v, q.g = f.master.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), f.current_grad().data_ptr()
    q.segs, q.unit_seg, q.unit_blk = dev.data_ptr(), unit_seg.data_ptr(), unit_blk.data_ptr()
    q.state, q.step_dev, q.lr_dev = state.data_ptr(), self._step_store.data_ptr(), self.lr_dev.data_ptr()
    q.lr, q.beta1, q.beta2, q.eps, q.weight_decay = float(self.lr), self.betas[0], self.betas[1], self.eps, self.weight_decay
    q.n_units, q.n_stages = len(seg_ids), n_stages
    for s_ in range(RIDER_STAGES + 1):  # entries of stage s: [stage_begin[s], stage_begin[s + 1]); beyond n_stages: the rest
      q.stage_begin[s_] = int(np.searchsorted(stages, s_, side='left')) if s_ <= n_stages else len(seg_ids)
    if chain is not None:
      if chain._queue is None:
        raise ValueError('chain: build the other optimizer\'s queue first')
      q.chain, q.chain_stages = chain._queue['dev'].data_ptr(), chain._queue['host'].n_stages
    qdev = torch.frombuffer(bytearray(bytes(q)), dtype=torch.uint8).clone().to(devc)
    self._queue = dict(host=q, dev=qdev, keep=(unit_seg, unit_blk, state, dev), state=state,
                       key=(self._seg_key, f.current_grad().data_ptr()), armed=False)
    return n_stages

  def queue_limit(self, stages):
    """Queue entries in the first `stages` stages (what riders may run once that many stages of the backward are done)."""
    q = self._queue['host']
    return int(q.stage_begin[max(0, min(int(stages), q.n_stages))])

  def queue_stats(self):
    """(queue entries, entries taken by riders over all finished steps, finished steps) -- synchronises."""
    from ._lib import RIDER_STAT0
    st = self._queue['state'][RIDER_STAT0:RIDER_STAT0 + 2].tolist()
    return self._queue['host'].n_units, st[0], st[1]

  def queue_ptr(self):
    return self._queue['dev'].data_ptr()

  def arm_queue(self, on=True):
    """From now on `step()` finishes the queue (mmt_adam_step_queue) instead of launching the fused kernel.  The caller
    guarantees that every backward between two steps carries the queue consistently (train_step.GraphedTrainStep)."""
    if self._queue is None:
      raise RuntimeError('arm_queue() without build_queue()')
    self._queue['armed'] = bool(on)

  def _queue_ok(self):
    """A queue exists and still describes the flat buffers (master, shadows and the gradient buffer in use)."""
    q = self._queue
    if q is None:
      return False
    self._segment_table()
    return q['key'] == (self._seg_key, self.flat.current_grad().data_ptr())

  def _queue_valid(self):
    q = self._queue
    return q is not None and q['armed'] and q['key'] == (self._seg_key, self.flat.current_grad().data_ptr())

  fuse_shadows = True
  _seg_key = _seg_table = None

  def _segment_table(self):
    """(host ctypes array, device copy, n) of the flat buffer's segments for mmt_adam_step_fused, rebuilt when the
    master or a shadow moved.  None: shadows not allocated yet / not refreshable -> plain step + lazy re-pack."""
    from ._lib import MmtAdamSeg
    f = self.flat
    key = (f.master.data_ptr(),) + tuple((sh['dst'].data_ptr() if sh['dst'] is not None else 0) for sh in f.shadows)
    if self._seg_key == key:
      return self._seg_table
    segs = f.adam_segments() if f.shadows else None
    table = None
    if segs is not None and 0 < len(segs) <= 160:
      arr = (MmtAdamSeg * len(segs))()
      for i, sg in enumerate(segs):
        it = arr[i]
        if sg[0] == 'plain':
          it.offset, it.count = sg[1], sg[2]
        else:
          sh = sg[2]
          it.offset, it.count = sg[1], sh['rows'] * sh['cols']
          it.dst = sh['dst'].data_ptr()
          it.dst_t = sh['dst_t'].data_ptr() if sh['transpose'] else None
          it.rows, it.cols, it.dst_ld = sh['rows'], sh['cols'], sh['dst_ld']
          it.dst_t_ld = sh['rows'] if sh['transpose'] else 0
      raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
      table = (arr, raw.to(f.master.device), len(segs))
    self._seg_key, self._seg_table = key, table
    return table


def build_optimizers(model, lr=5e-5, **kw):
  """(FlatAdam for the engine parameters, torch Adam for the rest or None)."""
  flat_ids = {id(p) for p in model.engine_params()}
  rest = [p for p in model.parameters() if p.requires_grad and id(p) not in flat_ids]
  return FlatAdam(model._flat, lr=lr, **kw), (torch.optim.Adam(rest, lr=lr, **kw) if rest else None)


def _reference_param_order(model):
  """The parameter list the reference hands its single optimizer: filter(requires_grad, model.parameters()), train.py:95-100."""
  return [p for p in model.parameters() if p.requires_grad]


def merged_state_dict(model, optimizers):
  """The state of several optimizers over disjoint subsets of `model`'s parameters (FlatAdam and/or torch.optim.Adam) as ONE
  torch.optim.Adam state dict over the reference's parameter order -- what base/base_trainer.py:353-365 would have saved
  for the same training run.  Parameters no optimizer holds state for (never stepped: the unused pooler) have no entry,
  exactly as torch leaves them out."""
  order = _reference_param_order(model)
  index = {id(p): i for i, p in enumerate(order)}
  state, group = {}, None
  for opt in optimizers:
    if opt is None:
      continue
    sd = opt.state_dict()
    params = [p for g in opt.param_groups for p in g['params']]
    ids = [i for g in sd['param_groups'] for i in g['params']]
    for local, p in zip(ids, params):
      st = sd['state'].get(local)
      if st is not None and id(p) in index:
        state[index[id(p)]] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
    if group is None:
      group = {k: v for k, v in sd['param_groups'][0].items() if k != 'params'}
      if torch.is_tensor(group.get('lr')):  # (a captured step keeps the rate of its torch Adam in a device scalar)
        group['lr'] = float(group['lr'])
  group = dict(group or {}, params=list(range(len(order))))
  return {'state': state, 'param_groups': [group]}


def load_merged_state_dict(model, optimizers, sd):
  """Inverse of merged_state_dict: a reference optimizer checkpoint (one Adam, reference parameter order;
  base/base_trainer.py:426-432) split over the optimizers of the step."""
  order = _reference_param_order(model)
  groups = sd['param_groups']
  if len(groups) != 1 or len(groups[0]['params']) != len(order):
    raise ValueError('expected ONE param group over the model\'s %d trainable parameters, got %s'
                     % (len(order), [len(g['params']) for g in groups]))
  ref_ids = groups[0]['params']
  at = {id(p): ref_ids[i] for i, p in enumerate(order)}
  hyper = {k: v for k, v in groups[0].items() if k != 'params'}
  for opt in optimizers:
    if opt is None:
      continue
    params = [p for g in opt.param_groups for p in g['params']]
    sub_state = {}
    for local, p in enumerate(params):
      st = sd['state'].get(at.get(id(p)), sd['state'].get(str(at.get(id(p)))))
      if st is not None:
        sub_state[local] = st
    own = opt.state_dict()['param_groups'][0]
    g = dict(own, **{k: v for k, v in hyper.items() if k in own and not torch.is_tensor(own[k])})
    g['params'] = list(range(len(params)))
    if isinstance(opt, torch.optim.Optimizer) and len(opt.state):
      # A torch optimizer that has stepped -- possibly inside a captured graph, which holds the ADDRESSES of its state
      # tensors: torch's load_state_dict would replace them (the graph would keep updating the orphans).  Copy in place.
      with torch.no_grad():
        for local, p in enumerate(params):
          cur, st = opt.state.get(p), sub_state.get(local)
          if cur is None:
            if st is not None:
              raise ValueError('optimizer state for a parameter this optimizer has never stepped: load before the first step')
            continue
          for k in ('step', 'exp_avg', 'exp_avg_sq'):
            if st is None:  # never stepped in the checkpoint: no moments of THIS run may survive
              if torch.is_tensor(cur[k]):
                cur[k].zero_()
              else:
                cur[k] = 0
            elif torch.is_tensor(cur[k]):
              src = torch.as_tensor(st[k])
              if src.numel() != cur[k].numel() or (src.dim() and cur[k].dim() and tuple(src.shape) != tuple(cur[k].shape)):
                raise ValueError("optimizer checkpoint: '%s' of parameter %d has shape %s, this optimizer holds %s"
                                 % (k, local, tuple(src.shape), tuple(cur[k].shape)))
              cur[k].copy_(src.to(cur[k].dtype).reshape(cur[k].shape))
            else:
              cur[k] = st[k]
        for k, v in g.items():
          if k != 'params' and not torch.is_tensor(opt.param_groups[0].get(k)):
            opt.param_groups[0][k] = v
      continue
    opt.load_state_dict({'state': sub_state, 'param_groups': [g]})
