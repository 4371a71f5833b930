"""One data-parallel training step of the hot path, captured as HIP graphs.

Eagerly, a step issues ~600 tiny launches (text heads, autograd glue, optimizer) and is host-bound at
~7 ms while the kernels need < 2 ms.  `GraphedTrainStep` captures the compute in HIP graphs and keeps the RCCL
collectives OUTSIDE of them (so nothing depends on collective capture support).  The basic multi-rank shape:

    graph A : CENet forward (out='embds') on the local batch
    eager   : ONE all-gather of the embeddings / weights over ranks (packed into a send buffer inside graph A)
    graph B : global similarity + loss + backward (through the gather, into the local embeddings,
              through graph A's autograd graph into the flat gradient buffer)
    eager   : all-reduce(SUM) of the flat gradient buffer(s) (+ a bucket for parameters outside them)
    graph C : optimizer step (fused flat Adam + capturable torch Adam for the rest)

On one rank A, B and C are captured as ONE graph (every graph boundary costs ~9 us).

With more than one rank the 68 MB gradient all-reduce would sit exposed between B and C (about a quarter of the step
on 8 GPUs over xGMI rings).  `overlap_grad_sync` (default: on when world size > 1) therefore cuts graph B at the
points where a contiguous span of the flat gradient buffer is final -- similarity/loss + text heads, then every
encoder layer from the top down, then embeddings + the expert projections -- and starts that span's all-reduce on
RCCL's stream while the next stage's graph runs:

    B0 loss, text heads, read-out, layer L-1 | all-reduce('top')   B1 layer L-2 | all-reduce   ...
    Bk layer 0 + embeddings + video tokens | all-reduce('bottom')   wait   C
(the flat layout is ordered back to front so that each of these spans is one contiguous run, CENet.grad_regions)

The stages call the engine's range backward (mmt_bert_backward_range) directly instead of through autograd, so each
stage is a plain kernel sequence that captures into its own graph; the collectives stay eager.

Fork mode (`fork` bits, opt-in -- measured slower on this runtime, see FORK_DEFAULT): the captured step is not one serial
chain of kernels.  Work that nothing on the critical
path waits for leaves the main stream for a second one, ordered by events that become EDGES of the captured graph:

    main: forward ............ loss | layer L-1 dgrad chain | layer L-2 dgrad chain | ... | layer 0 | embeddings, tokens |
    side: text heads fwd |            text heads bwd, dW(L-1), Adam(top) | dW(L-2), Adam(L-2) | ... | dW(0), dW(ReduceDim), Adam(bottom)

The weight gradients of a layer only feed the optimizer, so they run under the input-gradient chain of the layers below
(mmt_bert_backward_range, MmtBertBatch.fork); the optimizer runs region by region (`FlatAdam.step_span`) as soon as a
region's gradients are final -- HBM-bound work under MFMA-bound work; the text heads run beside the video tower.

Everything data-dependent lives in device memory (live row count of the token packing, dropout seed,
Adam step counter), so replays are correct for new minibatches copied into the static input buffers.
"""
import time
import torch
import torch.distributed as dist

import os

from . import dist as mdist
from . import _lib
from .feature_store import RaggedFeatures
from .model import cross_view_similarity
from .optim import FlatAdam


def _layer_stage(name, n_layers, first):
  """Optimizer-queue stage of an encoder parameter: `first` + k for the matrices and biases of layer n_layers-1-k (final
  after that layer's weight-gradient launch); None (final only when the backward ends) for LayerNorm parameters --
  their gradients come out of the batched reduction at the end of the backward --, embeddings and everything else."""
  import re
  z = re.search(r'encoder\.layer\.(\d+)\.', name)
  if z is None or '.layer_norm.' in name or '.LayerNorm.' in name:
    return None
  return first + (n_layers - 1 - int(z.group(1)))


_QUIESCE_WARNED = False


def _pending_nccl_works():
  """Collectives the NCCL/RCCL process group's watchdog still holds (issued, not yet retired), from the flight recorder;
  None when the recorder cannot tell (disabled, or this torch build has no such hook)."""
  import pickle
  try:
    from torch._C._distributed_c10d import _dump_nccl_trace
    if int(os.environ.get('TORCH_NCCL_TRACE_BUFFER_SIZE', '2000')) <= 0:
      return None
    active = pickle.loads(_dump_nccl_trace(True, False, True)).get('entries', [])
    if active:
      return len(active)
    # "nothing active" only means something if the recorder records at all: it must know the collectives issued so far
    seen = pickle.loads(_dump_nccl_trace(True, False, False)).get('entries', [])
    return 0 if seen else None
  except Exception:  # noqa: BLE001 -- any failure of the introspection hook: the caller falls back to the time-based wait
    return None


class FlatMinibatch(dict):
  """A minibatch (dict of tensors / dicts of tensors) laid out in ONE device buffer, so that loading it into the
  static input buffers of the captured graphs is a single device-to-device (or host-to-device) copy instead of
  ~45 tiny ones (the reference's move_dict_to_device, trainer/trainer.py:36-52, moves tensor by tensor)."""

  def __init__(self, minibatch, device, pin_memory=False):
    """device may be 'cpu' (+ pin_memory=True): a host-side staging buffer whose upload into a device FlatMinibatch of
    the same layout is one asynchronous H2D copy."""
    super().__init__()
    leaves = []
    self.ragged = []  # top-level keys holding RaggedFeatures

    def walk(d, out):
      for k, v in d.items():
        if isinstance(v, dict):
          out[k] = {}
          walk(v, out[k])
        elif torch.is_tensor(v):
          leaves.append((out, k, v))
        elif isinstance(v, RaggedFeatures):  # video features in the wire format: their own buffer, same idea
          out[k] = RaggedFeatures(v.layout, device, pin_memory).copy_from(v, non_blocking=False)
          self.ragged.append(k)
        else:
          out[k] = v

    walk(minibatch, self)
    off = 0
    spans = []
    for _, _, v in leaves:
      off = (off + 255) // 256 * 256
      spans.append(off)
      off += v.numel() * v.element_size()
    self.flat = torch.zeros(max(off, 1), dtype=torch.uint8, device=device,
                            pin_memory=bool(pin_memory) and torch.device(device).type == 'cpu')
    for (out, k, v), o in zip(leaves, spans):
      n = v.numel() * v.element_size()
      view = self.flat[o:o + n].view(v.dtype).view(v.shape)
      view.copy_(v)
      out[k] = view


def _clone_tree(src):
  """A second set of input buffers with the layout of `src` (FlatMinibatch, or a plain dict of device tensors)."""
  if isinstance(src, FlatMinibatch):
    return FlatMinibatch(src, src.flat.device)
  out = {}
  for k, v in src.items():
    if isinstance(v, dict):
      out[k] = _clone_tree(v)
    elif torch.is_tensor(v):
      out[k] = v.clone()
    elif isinstance(v, RaggedFeatures):
      out[k] = RaggedFeatures(v.layout, v.device).copy_from(v, non_blocking=False)
    else:
      out[k] = v
  return out


def _copy_tree(dst, src):
  for k, v in src.items():
    if isinstance(v, dict):
      _copy_tree(dst[k], v)
    elif torch.is_tensor(v):
      dst[k].copy_(v, non_blocking=True)
    elif isinstance(v, RaggedFeatures):
      dst[k].copy_from(v)


# GraphedTrainStep(fork=...) bits: 1 | 2 | 4 are the engine's _lib.FORK_WGRAD / FORK_EARLY / FORK_REDUCE
FORK_WGRAD, FORK_EARLY, FORK_REDUCE = _lib.FORK_WGRAD, _lib.FORK_EARLY, _lib.FORK_REDUCE
FORK_ADAM = 16    # optimizer region by region on the side stream
FORK_TEXT = 32    # text heads (forward and backward) on the side stream
FORK_TOKENS = 64  # ReduceDim weight gradients on the side stream
FORK_ALL = FORK_WGRAD | FORK_REDUCE | FORK_ADAM | FORK_TEXT | FORK_TOKENS
# Measured on MI355X / ROCm 7.2 (profiles/r03_fork_lab.txt): every variant is SLOWER than the serial chain (1.465 ms ->
# 1.51-1.59 ms).  The graph executor maps the branches to separate hardware queues; each cross-queue edge costs 5-15 us,
# and kernels that do overlap slow each other down by as much as they overlap (a 16 us input-gradient GEMM takes 65-73 us
# next to the 256-tile weight-gradient launch, which itself goes from 57 to 72-103 us).  So the default is one chain.
FORK_DEFAULT = 0


class GraphedTrainStep:

  def __init__(self, model, loss_fn, minibatch, lr=5e-5, group=None, use_graphs=True, warmup_steps=3,
               overlap_grad_sync=None, force_collectives=False, grad_dtype=None, capture_collectives=False, fork=None,
               grad_algo='allreduce', split_bottom=True, input_slots=1, bind_inputs=None, shard_optimizer=False,
               host_feed=None, adam_riders=None, live_rows=None):
    """minibatch: dict of DEVICE tensors as CENet.forward takes them (used as the static input buffers).
    overlap_grad_sync: None = staged backward with per-stage all-reduce when world size > 1; True forces the staged
    backward (also at world size 1, where it only splits graph B); False = one all-reduce after the backward.
    grad_dtype: torch.bfloat16 sends the gradient all-reduces in bf16 (half the bytes over xGMI, `dist.WireBuffer`);
    None / torch.float32 reduces the fp32 buffer in place.
    grad_algo: 'allreduce' | 'rs_ag' (reduce-scatter + all-gather per span, `dist.WireBuffer`).
    shard_optimizer (with grad_algo='rs_ag', world size > 1): each rank runs Adam on the 1/N shard of every span whose
    reduced gradients the reduce-scatter left with it, and the UPDATED WEIGHTS are all-gathered instead of the reduced
    gradients (same bytes on the wire) -- the optimizer pass (HBM-bound: 28 B per parameter) shrinks N-fold per rank,
    at the price of one re-pack of the bf16 weight shadows.  Bit-identical weights to the all-reduce path
    (tests/test_dist_cpu.py, tests/test_dp_gpu.py).  Reference: train.py:97-103, trainer/trainer.py:203-204.
    split_bottom: staged mode reduces layer 0's gradients before the embedding / token stage runs, so that only the
    expert projections + embedding tables (about half of the last span) are reduced after the backward has ended.
    input_slots: K > 1 keeps K sets of static input buffers and captures the step once per set (same kernels, same
    weights / optimizer state / seeds; only the input pointers differ): `step(slot)` then runs on whatever was put into
    `inputs(slot)` -- a loader writes minibatch i + 1 straight into the next slot (`upload`, `load(mb, slot)`) while step i
    runs, and the device-to-device copy of the whole minibatch into ONE set of static buffers (12 us for 26 MB at
    config B) disappears from the step.  bind_inputs(static): called before the captures of each slot, for modules
    that hold a pointer to an input tensor (bench.py's synthetic text tower).
    host_feed: list of K = input_slots PINNED host FlatMinibatches, one per slot (where a loader deposits minibatches,
    trainer/trainer.py:36-52,167): the captured step of slot s then CONTAINS the upload of slot (s + 1) % K from its pinned
    buffer -- a host-to-device copy node with no predecessor inside the graph, on a branch of its own, joined at the end.
    No cross-stream event is recorded or waited for per step (each cost 35-70 us of queue plumbing on this runtime: the
    r03 upload path ran 0.16 ms per step behind the resident one whatever the bytes); `prime()` uploads slot 0 once.
    Contract for the loader: pinned buffer (s + 1) % K holds minibatch i + 1 when step(s) is launched for minibatch i and is
    not rewritten before that step has finished (`step_done(slot)`).
    live_rows: packed token rows of the minibatches as the loader counts them (`CENet.count_live_rows`; RaggedFeatures
    carry the count themselves): the GEMM dispatcher picks its tiles for a launch's LIVE size (include/mmt_hip.h:
    MmtBertBatch.live_rows_hint) at capture time.  `step(slot, live_rows=n)` re-captures (once per distinct tile choice, kept)
    when a minibatch's count would select other tiles -- on ONE rank; with several ranks the constructor's count stands (a
    re-capture contains collectives that all ranks would have to enter); without any count a packed batch is priced at its
    allocated rows.
    adam_riders (one rank only; None = OFF unless MMT_ADAM_RIDERS=1): the optimizer inside the backward -- the step's Adam
    update is a queue of 4096-element units ordered by when their gradients are final, the GEMM launches of the
    backward carry it (blocks without a tile of their own -- idle CUs, the last partial round -- stream Adam's bytes beside
    the tiles) and the optimizer launch at the end only runs what is left.  Bit-identical to the serial fused step
    (tests/test_optim_gpu.py); needs the stage-by-stage backward (native text heads and losses).  MEASURED SLOWER on
    MI355X in every configuration (r06, DESIGN section 7: the hosting GEMMs are bound by memory latency, every unit a rider
    streams costs 2-3x what it costs in the optimizer's own launch -- headline 1.31-1.58 ms against 1.28 ms), hence opt-in.
    Reference: train.py:100, trainer/trainer.py:203-204.
    The warm-up steps only allocate buffers and optimizer state: weights, Adam moments and step count, BatchNorm
    statistics and the dropout seed are restored afterwards, so the first `step()` IS the first optimisation step."""
    self.model, self.loss_fn, self.group = model, loss_fn, group
    if fork is None:
      fork = int(os.environ.get('MMT_FORK', FORK_DEFAULT))
    self.fork = int(fork)
    self._side = None
    self._fork_on = False  # decided after the first warm-up step (needs the model's stage handles)
    if adam_riders is None:
      adam_riders = os.environ.get('MMT_ADAM_RIDERS', '0') == '1'
    self._want_riders = bool(adam_riders)
    self._rider_on = False  # decided after the first warm-up step as well
    self._keep = []
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.rank = dist.get_rank(group) if dist.is_initialized() else 0
    if live_rows is not None and hasattr(model, 'live_rows_hint'):
      model.live_rows_hint = int(live_rows)
    self._sig, self._by_sig, self._sig_cache = None, {}, {}
    # measurement hook: issue the collectives even at world size 1 (a 1-rank RCCL group) to see what their stream
    # plumbing costs on a single GPU (bench.py --force-collectives)
    self._force_coll = bool(force_collectives) and dist.is_initialized()
    self._multi = self.world > 1 or self._force_coll
    self._want_stages = self._multi if overlap_grad_sync is None else bool(overlap_grad_sync)
    self.staged = False
    self.static = minibatch
    self._statics = [minibatch]
    self._bind = bind_inputs
    self._caps = None
    self._slot_ev = None
    self._host_feed = list(host_feed) if host_feed else None
    self._feed_stream = None
    self._next_upload = None  # (static, pinned) the capture in progress should carry
    self._done_ev = None
    self.capture_collectives = bool(capture_collectives)
    # With a process group alive, its watchdog thread polls events of collectives still in flight (the eager all-gather
    # issued between two captures, a neighbour's slow broadcast): in 'global' mode such a call from ANOTHER thread
    # invalidates a capture that happens to be open.  Calls of the capturing thread itself stay checked.
    self._cap_mode = 'thread_local' if dist.is_initialized() else 'global'
    self._one_graph = False
    self._pool = None  # graph memory pool shared by the captures of every input slot
    self._staging = None  # device-side landing buffer of prefetch()
    flats = model.flats() if hasattr(model, 'flats') else [model._flat]  # video side (+ the native text tower's)
    flat_ids = {id(p) for f in flats for p in f.params}
    rest = [p for p in model.parameters() if p.requires_grad and id(p) not in flat_ids]
    self.opt_flats = [FlatAdam(f, lr=lr) for f in flats]
    self.opt_flat = self.opt_flats[0]
    # parameters outside the flat buffers (a foreign text tower, txt_pro='lin' heads): stock Adam whose learning rate
    # lives in a DEVICE tensor when the step is captured -- a python float would be frozen into the graph and the
    # schedule (set_lr) would silently stop reaching these parameters
    self._rest_lr = None
    if rest:
      dev0 = rest[0].device
      if use_graphs and dev0.type == 'cuda':
        self._rest_lr = torch.tensor(float(lr), device=dev0, dtype=torch.float32)
      self.opt_rest = torch.optim.Adam(rest, lr=self._rest_lr if self._rest_lr is not None else lr, capturable=use_graphs)
    else:
      self.opt_rest = None
    self.grad_dtype = grad_dtype
    self.grad_algo, self._split_bottom = grad_algo, bool(split_bottom)
    self.shard_opt = bool(shard_optimizer) and self.world > 1
    if shard_optimizer and grad_algo != 'rs_ag':
      raise ValueError("shard_optimizer needs grad_algo='rs_ag' (the shards are what the reduce-scatter leaves on a rank)")
    self._wire = mdist.WireBuffer(grad_dtype, grad_algo)
    self.syncs = [mdist.GradSync(f, rest if i == 0 else (), group, grad_dtype=grad_dtype, algo=grad_algo)
                  for i, f in enumerate(flats)]
    # sharded optimizer: flat buffer -> its optimizer, the whole-buffer spans of the un-staged step, the all-reduce
    # bucket of the parameters outside the flat buffers -- built once (they used to be rebuilt on every step), and the
    # (flat, offset, count) spans the shards of the last step were cut from (optimizer_state_dict gathers along them)
    self._opt_of = {id(o.flat): o for o in self.opt_flats}
    self._whole_regions = {'flat%d' % i: (o.flat, 0, o.flat.count) for i, o in enumerate(self.opt_flats)}
    self._rest_sync = mdist.GradSync(None, rest, group) if self.shard_opt else None
    self._shard_spans = []
    self._exposed = []  # (event before, event after) around the final wait for the staged reductions, last steps
    self.sync = self.syncs[0]
    self._extra_flats = flats[1:]
    self.use_graphs = use_graphs
    self.loss = None
    self._graphs = None
    # Warm-up AND capture run on one dedicated side stream: autograd's AccumulateGrad nodes remember the
    # stream they were created on, and a node bound to the default stream breaks capture of the backward.
    self._stream = torch.cuda.Stream()
    self._stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(self._stream):
      snap = self._snapshot() if warmup_steps > 0 else None
      for i in range(warmup_steps):  # allocates every lazily created buffer / optimizer state
        self._eager_step()
        if i == 0 and self._want_riders and not self._multi and not self.fork and not self._want_stages:
          self._rider_on = self._stageable([p for p in rest if p.grad is not None]) and \
              all(not o._frozen_spans() for o in self.opt_flats)
        if i == 0 and (self._want_stages or self.fork):
          ok = self._stageable([p for p in rest if p.grad is not None])  # e.g. the unused pooler: no grad
          self.staged = ok and self._want_stages
          if not ok or len(flats) > 1 or not hasattr(model, '_text_heads_backward'):
            self.fork = 0  # fork mode drives the backward stage by stage, like the staged reduction
          if self.fork:
            self._side = torch.cuda.Stream()
            if self.fork & FORK_TEXT:
              model.overlap_text_heads = True
              model._side_streams[next(model.parameters()).device] = self._side
            self._fork_on = True
      if self._rider_on:  # the queues exist before anything is captured (building them copies tables to the device)
        self._arm_riders()
        self._disarm_riders()
      if snap is not None:
        self._restore(snap)
    torch.cuda.current_stream().wait_stream(self._stream)
    if use_graphs:
      if int(input_slots) > 1:
        self._statics += [_clone_tree(minibatch) for _ in range(int(input_slots) - 1)]
        if self._host_feed is not None and len(self._host_feed) != len(self._statics):
          raise ValueError('host_feed: one pinned FlatMinibatch per input slot')
      self._capture_slots()
      self._sig = self._tile_signature(getattr(model, 'live_rows_hint', None))

  def _capture_slots(self):
    """The captured step for every input slot, under the model's current live-row hint (= tile choice)."""
    self._n_capture_sets = getattr(self, '_n_capture_sets', 0) + 1
    if len(self._statics) > 1:
      self._caps = []
      for si, st in enumerate(self._statics):
        self.static = st
        if self._host_feed is not None:
          nx = (si + 1) % len(self._statics)
          self._next_upload = (self._statics[nx], self._host_feed[nx])
        if self._bind:
          self._bind(st)
        # the slots never replay concurrently: their captures share ONE graph memory pool (activations / workspaces of a
        # step exist once, not once per slot); what a capture leaves alive (its loss / embedding outputs) stays allocated
        self._capture()
        if self._pool is None:
          g0 = self._graphs[0]
          self._pool = g0.pool()
        self._caps.append((self._graphs, self._e, self.loss, self._one_graph))
      self.static = self._statics[0]
      if self._bind:
        self._bind(self.static)
      self._graphs, self._e, self.loss, self._one_graph = self._caps[0]
    else:
      self._capture()

  def _tile_signature(self, live_rows):
    """The tiles the dispatcher picks for the video encoder's GEMMs at `live_rows` packed token rows (None: not applicable).
    Two minibatches with the same signature replay the same captured step; another signature is another capture."""
    m = self.model
    vb = getattr(m, 'vid_bert', None)
    if live_rows is None or vb is None or not getattr(m, 'pack_tokens', False) or not getattr(m, '_plans', None):
      return None
    hit = self._sig_cache.get(int(live_rows))
    if hit is not None:
      return hit
    L, E = _lib.lib(), _lib.EPI
    rows = next(iter(m._plans.values())).rows
    d, I = vb.config.hidden_size, vb.config.intermediate_size
    shapes = [(E['BIAS_BF16'], 3 * d, d, 0), (E['BIAS_DROP_RES'], d, d, 0), (E['BIAS_GELU'], I, d, 0), (E['BIAS_DROP_RES'], d, I, 0),
              (E['DGELU'], I, d, 0), (E['ADD_F32'], d, I, 0), (E['BF16'], d, d, 1), (E['ADD_F32'], d, 3 * d, 0)]
    sig = tuple(L.mmt_gemm_select_tile(e, rows, n, k, 1, int(live_rows), 0, dot, 0) for e, n, k, dot in shapes)
    self._sig_cache[int(live_rows)] = sig
    return sig

  def _switch_tiles(self, sig, live_rows):
    """step(live_rows=...) met a minibatch whose live size selects other GEMM tiles than the captures in use: keep those,
    take (or capture, once) the set for the new choice."""
    self._by_sig[self._sig] = (self._caps, self._graphs, self._e, self.loss, self._one_graph)
    self.model.live_rows_hint = int(live_rows)
    hit = self._by_sig.get(sig)
    if hit is not None:
      self._caps, self._graphs, self._e, self.loss, self._one_graph = hit
    else:
      cur = torch.cuda.current_stream()
      self._stream.wait_stream(cur)
      with torch.cuda.stream(self._stream):
        self._capture_slots()
      cur.wait_stream(self._stream)
    self._sig = sig

  # ---- warm-up must not train --------------------------------------------------------------------
  def _snapshot(self):
    """Everything a training step mutates, captured before the warm-up steps."""
    m = self.model
    snap = dict(buffers=[(b, b.detach().clone()) for b in m.buffers()], flats=[], rest=[], seeds=[])
    for f in (m.flats() if hasattr(m, 'flats') else [m._flat]):
      if f.master is None or not f.is_flat():
        f.ensure(f.params[0].device)  # what the first forward would do
      snap['flats'].append((f, f.master.detach().clone()))
    flat_ids = {id(p) for f, _ in snap['flats'] for p in f.params}
    snap['rest'] = [(p, p.detach().clone()) for p in m.parameters() if id(p) not in flat_ids]
    for mod in m.modules():
      sd = getattr(mod, '_seed_dev', None)
      if torch.is_tensor(sd):
        snap['seeds'].append((mod, sd.detach().clone()))
    return snap  # (a dropout seed first drawn during the warm-up simply starts its stream a few values later)

  @torch.no_grad()
  def _restore(self, snap):
    for b, v in snap['buffers']:
      b.copy_(v)
    for f, v in snap['flats']:
      f.master.copy_(v)
      f._dirty = True
      f.pack()  # the bf16 shadows follow the restored weights NOW (not inside the graph that is captured next)
    for p, v in snap['rest']:
      p.copy_(v)
    for mod, v in snap['seeds']:
      mod._seed_dev.copy_(v)
    for o in self.opt_flats:  # Adam state as before the first step: zero moments, step 0 (buffers stay allocated)
      if o.exp_avg is not None:
        o.exp_avg.zero_()
        o.exp_avg_sq.zero_()
        o.step_dev.zero_()
      if o._queue is not None:
        o._queue['state'].zero_()
    if self.opt_rest is not None:
      for st in self.opt_rest.state.values():
        for v in st.values():
          if torch.is_tensor(v):
            v.zero_()
    self._zero()

  # ---- pieces ------------------------------------------------------------------------------------
  def _upload_branch_begin(self):
    """Inside a capture: the next slot's host-to-device copy as a ROOT node of the graph (a branch of its own)."""
    if self._next_upload is None or not torch.cuda.is_current_stream_capturing():
      return
    dst, src = self._next_upload
    if self._feed_stream is None:
      self._feed_stream = torch.cuda.Stream()
    self._feed_stream.wait_stream(torch.cuda.current_stream())  # joins the capture; nothing captured yet: no predecessor
    with torch.cuda.stream(self._feed_stream):
      dst.flat.copy_(src.flat, non_blocking=True)

  def _upload_branch_end(self):
    if self._next_upload is None or not torch.cuda.is_current_stream_capturing():
      return
    torch.cuda.current_stream().wait_stream(self._feed_stream)

  def prime(self, slot=0):
    """host_feed mode: bring `slot`'s pinned minibatch into its input buffers before the first step (synchronous)."""
    if self._host_feed is None:
      raise RuntimeError('prime() needs host_feed')
    self._statics[slot].flat.copy_(self._host_feed[slot].flat, non_blocking=False)
    torch.cuda.synchronize()

  def step_done(self, slot):
    """host_feed mode: host-side wait until the last step(slot) has finished -- its graph carried the upload of slot
    (slot + 1) % K, whose pinned buffer may be refilled from then on."""
    if self._done_ev is not None and self._done_ev[slot] is not None:
      self._done_ev[slot].synchronize()

  def _forward(self):
    self._upload_branch_begin()
    mb = self.static
    e = self.model(mb['token_ids'], mb['features'], mb.get('features_t'), mb.get('features_ind'),
                   mb.get('features_avgpool'), mb.get('features_maxpool'), mb['query_masks'], out='embds')
    self._pack_local(e)
    return e

  _send = _recv = _gkeys = None

  def _pack_local(self, e):
    """The local embeddings / weights laid out in ONE send buffer (runs inside graph A): one all-gather per step
    instead of four -- every collective costs ~25 us of stream plumbing here, more than these transfers."""
    if not self._multi:
      return
    if self._send is None:
      self._gkeys = [(k, tuple(e[k].shape), e[k].numel()) for k in sorted(e)]
      total = sum(n for _, _, n in self._gkeys)
      dev = e[self._gkeys[0][0]].device
      self._send = torch.empty(total, device=dev, dtype=torch.float32)
      self._recv = torch.empty(self.world * total, device=dev, dtype=torch.float32)
    off = 0
    for k, _, n in self._gkeys:
      self._send[off:off + n].copy_(e[k].detach().reshape(-1))
      off += n

  def _gather(self, e):
    """-> dict of the global batch as [world, numel] strided views of the receive buffer (see _globalize)."""
    if not self._multi:
      return {k: v.detach() for k, v in e.items()}
    dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
    recv, out, off = self._recv.view(self.world, -1), {}, 0
    for k, _, n in self._gkeys:
      out[k] = recv[:, off:off + n]
      off += n
    return out

  def _globalize(self, e, g):
    """[world, numel] views -> contiguous global-batch tensors shaped like the local ones (copies: inside graph B)."""
    if not self._multi:
      return g
    shapes = {k: shape for k, shape, _ in self._gkeys}
    return {k: v.reshape((self.world * shapes[k][0],) + shapes[k][1:]) for k, v in g.items()}

  def _loss_backward(self, e, g):
    g = self._globalize(e, g)
    fast = self._fast_loss_grads(e, g, fuse_readout=True)
    if fast is not None:
      loss, outs, grads = fast
      torch.autograd.backward(outs, grads)
      return loss
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

  def _fast_loss_grads(self, e, g, fuse_readout=False):
    """Our own similarity + loss kernels called directly (no autograd bookkeeping for this tiny sub-graph: saves the
    ones_like / multiply / slice launches): global sims -> loss + dL/dsims in one kernel -> similarity backward ->
    (loss, [local outputs of graph A that need a gradient], [their gradients]).  None = not applicable (foreign loss
    module, several captions per video): the generic autograd path is used."""
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
    b = e['vid_embds'].shape[0]
    sl = slice(self.rank * b, (self.rank + 1) * b)
    if 2 <= n <= L.mmt_simloss_small_max_n():  # (n == 1: the generic kernels below; the fused one rejects it)
      # small (single-rank) batches: loss, d loss / d sims and the whole similarity backward in ONE launch -- and, when
      # the video embeddings are the read-out of this model's own encoder output, the read-out backward too
      kind = 0 if isinstance(self.loss_fn, MaxMarginRankingLoss) else 1
      margin = float(getattr(self.loss_fn, 'margin', 0.0))
      fix_norm = int(getattr(self.loss_fn, 'fix_norm', True))
      st = getattr(self.model, '_stages', None) if (n == b and fuse_readout) else None
      fused = st is not None and st.get('vid_embds') is e['vid_embds'] and st.get('readout_inv') is not None
      dtxt, dtw = torch.empty_like(txt), torch.empty_like(tw)
      dvid = dlast = None
      if fused:
        last = st['last']
        dlast = (torch.empty if st['readout_compact'] else torch.zeros)(last.shape, device=dev, dtype=torch.float32)
      else:
        dvid = torch.empty_like(vid)
      check(L.mmt_simloss_bwd_small(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), ops._p(sims), ops._p(dots), n, m, d, kind,
                                    margin, fix_norm, ops._p(loss), ops._p(dtxt), ops._p(dvid), ops._p(dtw), None,
                                    ops._p(st['readout_inv']) if fused else None,
                                    ops._p(st['readout_rows']) if fused and st['readout_rows'] is not None else None,
                                    ops._p(dlast), ops._stream()), 'mmt_simloss_bwd_small')
      outs, grads = [], []
      pairs = [('text_embds', dtxt.view(n, m, 1, d)), ('text_weights', dtw.view(n, 1, m))]
      if fused:
        outs.append(st['last'])
        grads.append(dlast)
      else:
        pairs.insert(0, ('vid_embds', dvid))
      for k, gfull in pairs:
        if e[k].requires_grad:
          outs.append(e[k])
          grads.append(gfull[sl])
      return loss, outs, grads
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
    outs, grads = [], []
    for k, gfull in (('vid_embds', dvid), ('text_embds', dtxt.view(n, m, 1, d)), ('text_weights', dtw.view(n, 1, m))):
      if e[k].requires_grad:
        outs.append(e[k])
        grads.append(gfull[sl])
    return loss, outs, grads

  # ---- staged backward (gradient all-reduce overlapped with the remaining backward) -----------------
  def _stageable(self, rest):
    m = self.model
    st = getattr(m, '_stages', None)
    if rest or st is None or not getattr(m, '_native_text_heads', False):
      return False
    from .loss import InfoNceLoss, MaxMarginRankingLoss
    if not isinstance(self.loss_fn, (MaxMarginRankingLoss, InfoNceLoss)):
      return False
    return all(p.requires_grad for p in m._flat.params)

  def _stage_list(self, e, g, single_range=False):
    """[(callable, [names of the flat-gradient regions that are final once it has run])], in execution order.
    single_range: ONE stage -- loss, text heads, read-out, then the whole encoder as one engine call (one batched
    LayerNorm / table reduction at its end instead of one per stage) and the video tokens."""
    model, vb = self.model, self.model.vid_bert
    st = {}

    def head():
      fast = self._fast_loss_grads(e, self._globalize(e, g), fuse_readout=True)
      if fast is None:
        raise RuntimeError('staged backward needs one caption per video and the native losses')
      self.loss, outs, grads = fast
      pairs = list(zip(outs, grads))
      h = model._stages
      txt = [(o, gr) for o, gr in pairs if o is not e['vid_embds'] and o is not h['last']]
      if txt and not self._text_heads_backward_forked(e, txt):
        torch.autograd.backward([o for o, _ in txt], [gr for _, gr in txt])  # text heads
      dlast = next((gr for o, gr in pairs if o is h['last']), None)  # the fused kernel already ran the read-out backward
      if dlast is None:
        gvid = next(gr for o, gr in pairs if o is e['vid_embds'])
        dlast, = torch.autograd.grad([e['vid_embds']], [h['last']], [gvid])  # read-out backward only
      if self._fork_on:
        # weight gradients / reductions of the encoder leave the main stream (MmtBertBatch.fork); between separately
        # captured stage graphs (multi-rank) every range call joins before it returns
        h['batch'].side_stream = self._side
        h['batch'].fork = (self.fork & (FORK_WGRAD | FORK_EARLY | FORK_REDUCE)) | (_lib.FORK_JOIN if self._multi else 0)
      st['run'] = vb.backward_ranges(h['batch'], dlast, vb.training)

    def bottom():
      st['run'](0, 0)
      side = self._side if (self._fork_on and self.fork & FORK_TOKENS) else None
      model._video_tokens_backward(model._stages['plan'], st['run'].dfeat, side_stream=side)
      if side is not None and self._multi:
        self._join()

    n_layers = vb.config.num_hidden_layers

    def top():
      head()
      if n_layers >= 2:
        st['run'](n_layers - 1, n_layers - 1)

    names = dict(self._grad_regions())
    if single_range:
      def whole():
        head()
        st['run'](n_layers - 1, 0)
        model._video_tokens_backward(model._stages['plan'], st['run'].dfeat)
      return [(whole, list(names) + ['flat%d' % (i + 1) for i in range(len(self._extra_flats))])]
    stages = [(top, (['top'] if 'top' in names else []) + ['flat%d' % (i + 1) for i in range(len(self._extra_flats))])]
    for l in range(n_layers - 2, 0, -1):
      stages.append((lambda l=l: st['run'](l, l), ['layer%d' % l]))
    if 'layer0' in names:  # split bottom: layer 0 alone, then the embedding stage + video tokens
      def bottom_split():
        st['run'](-1, -1)
        side = self._side if (self._fork_on and self.fork & FORK_TOKENS) else None
        model._video_tokens_backward(model._stages['plan'], st['run'].dfeat, side_stream=side)
        if side is not None and self._multi:
          self._join()
      stages.append((lambda: st['run'](0, 0, embed=False), ['layer0']))
      stages.append((bottom_split, ['bottom']))
    else:
      stages.append((bottom, ['bottom']))
    return stages

  def _grad_regions(self):
    """The model's flat-gradient spans in backward order; in staged multi-rank mode with layer 0 as its own span."""
    return self.model.grad_regions(split_bottom=self._split_bottom and self.staged)

  def _text_heads_backward_forked(self, e, txt):
    """Fork mode: the text heads' backward (3 launches, latency-bound) on the side stream, called directly -- autograd
    would run it on the stream of its forward too, but joins that stream into the caller's right after.  Only when
    nothing upstream of the heads needs a gradient (a trainable text tower takes the autograd path).  -> done?"""
    m = self.model
    if not (self._fork_on and self.fork & FORK_TEXT) or self._multi:
      return False
    text = getattr(m, '_th_text', None)
    fn = getattr(e['text_embds'], 'grad_fn', None)
    if text is None or fn is None or any(t is not None and t.requires_grad for t in (text, m._th_text_moe)):
      return False
    if getattr(m, '_th_needs_input_grad', True):
      return False
    grads = {id(o): gr for o, gr in txt}
    de, dtw = grads.get(id(e['text_embds'])), grads.get(id(e['text_weights']))
    if de is None:
      return False
    cur = torch.cuda.current_stream()
    self._side.wait_stream(cur)
    self._keep += [de, dtw]  # allocated on `cur`, read on the side stream: alive until the step has joined
    with torch.cuda.stream(self._side):
      m._text_heads_backward(e['text_embds'].shape[2], de, dtw, False, False, m.training)
    return True

  def _join(self):
    """main stream waits for the side stream (graph edge under capture)."""
    if self._side is not None:
      torch.cuda.current_stream().wait_stream(self._side)

  def _adam_regions(self, names, last):
    """Fork mode, one rank: the optimizer over the regions a stage has just finished, on the side stream -- after that
    stage's forked weight gradients (same stream) and everything the main stream has issued so far."""
    cur = torch.cuda.current_stream()
    self._side.wait_stream(cur)
    with torch.cuda.stream(self._side):
      for i, n in enumerate(names):
        flat, off, cnt = self._regions[n]
        self.opt_flat.step_span(off, cnt, bump=last and i == len(names) - 1)

  def _region_table(self):
    """name -> (flat, offset, count): the video flat's spans in backward order + every other flat as one span (the
    native text tower's backward runs with the text heads, in the first stage)."""
    tab = {n: (self.model._flat, off, cnt) for n, (off, cnt) in self._grad_regions()}
    for i, f in enumerate(self._extra_flats):
      tab['flat%d' % (i + 1)] = (f, 0, f.count)
    return tab

  def _reduce_async(self, names):
    if not self._multi:
      return []
    out = []
    for n in names:
      flat, off, cnt = self._regions[n]
      span = flat.current_grad()[off:off + cnt]
      if self.shard_opt:
        work, shard, _, lo, own = self._wire.reduce_scatter(span, self.group)
        out.append((work, dict(flat=flat, off=off, cnt=cnt, span=span, shard=shard, lo=lo, own=own)))
      else:
        out.append(self._wire.reduce(span, self.group))
    return out

  def _finish(self, handles):
    """handles: [(work, finish)] of WireBuffer.reduce -- wait for every collective, then unpack the wire buffers.  Sharded
    optimizer: [(work, span record)] of WireBuffer.reduce_scatter -- Adam on this rank's shard of every span, all-gather of
    the updated weights, one re-pack of the bf16 shadows."""
    for h, _ in handles:
      h.wait()
    if not self.shard_opt:
      for _, fin in handles:
        fin()
      return
    seen = set()
    self._shard_spans = [(r['flat'], r['off'], r['cnt']) for _, r in handles]
    self._moments_stale = True  # (until gather_optimizer_state())
    for _, r in handles:
      opt = self._opt_of[id(r['flat'])]
      grad = self._wire.shard_f32(r['span'], r['shard'])
      opt.step_shard(r['off'] + r['lo'], r['own'], grad, first=id(opt) not in seen)
      seen.add(id(opt))
    gathers = [self._wire.all_gather_span(r['flat'].master[r['off']:r['off'] + r['cnt']], self.group) for _, r in handles]
    for w, _ in gathers:
      w.wait()
    for _, fin in gathers:
      fin()
    for o in self.opt_flats:
      if id(o) in seen:
        o.flat.pack(force=True)

  def _zero(self):
    for o in self.opt_flats:
      o.zero_grad()
    if self.opt_rest is not None:
      self.opt_rest.zero_grad(set_to_none=True)

  def _opt(self):
    if not self.shard_opt:  # (sharded: the flat buffers were stepped shard by shard in _finish)
      for o in self.opt_flats:
        o.step()
    if self.opt_rest is not None:
      self.opt_rest.step()

  def _sync_all(self):
    if self.shard_opt:  # every flat buffer as ONE span; parameters outside them keep the all-reduce bucket
      # NOTE: in shard mode the flat gradient buffers are never reduced as a whole -- after the step a rank holds the
      # global sum only inside the shard buffers of `_wire`; flat.current_grad() keeps the rank-LOCAL gradients (anything
      # that reads gradients after the step -- norm logging, clipping -- must not assume they are global).
      self._regions = self._whole_regions
      for o in self.opt_flats:
        mdist.gather_stray_grads(o.flat)
      self._finish(self._reduce_async(list(self._regions)))
      self._rest_sync.sync(force=self._force_coll)
      return
    for sy in self.syncs:
      sy.sync(force=self._force_coll)

  def optimizer_state_dict(self):
    """The optimizer state of the step as the REFERENCE would have checkpointed it: one torch.optim.Adam state dict over
    filter(requires_grad, model.parameters()) (train.py:95-100, base/base_trainer.py:353-365).  LOCAL (no collective): any
    one rank may call it, as the reference's rank-0-only `_save_checkpoint` does.  Under `shard_optimizer=True` a rank
    holds current Adam moments for its own shards only, so every rank must have called `gather_optimizer_state()` since
    the last step -- otherwise this raises instead of pairing a non-zero step count with stale moments (or hanging in a
    collective the other ranks never enter)."""
    from .optim import merged_state_dict
    if self.shard_opt and self._multi and self._moments_stale:
      raise RuntimeError('GraphedTrainStep.optimizer_state_dict(): the Adam moments are sharded over the ranks '
                         '(shard_optimizer=True) and have not been gathered since the last step -- call '
                         'gather_optimizer_state() on EVERY rank first (a collective), then optimizer_state_dict() on the '
                         'rank that writes the checkpoint')
    return merged_state_dict(self.model, self.opt_flats + [self.opt_rest])

  _moments_stale = False

  def gather_optimizer_state(self):
    """COLLECTIVE (every rank of the group calls it; a no-op without the sharded optimizer): a rank has updated exp_avg /
    exp_avg_sq only inside ITS shard of every span, so a checkpoint written from one rank's buffers would pair a non-zero
    step count with stale moments for (N-1)/N of the parameters.  All-gather the moment shards along the geometry the last
    step cut them with (`WireBuffer.all_gather_span`: the same rank * per offsets as the reduce-scatter), through
    temporaries that are freed again.  Afterwards `optimizer_state_dict()` is local on every rank until the next step."""
    if not (self.shard_opt and self._multi) or not self._shard_spans:
      self._moments_stale = False
      return
    with torch.no_grad():
      for flat, off, cnt in self._shard_spans:
        opt = self._opt_of[id(flat)]
        if opt.exp_avg is None:
          continue
        for buf in (opt.exp_avg, opt.exp_avg_sq):
          work, fin = self._wire.all_gather_span(buf[off:off + cnt], self.group, async_op=False, scratch=True)
          if work is not None:
            work.wait()
          fin()
    self._moments_stale = False

  def load_optimizer_state_dict(self, sd):
    """Resume from a reference optimizer checkpoint (base/base_trainer.py:426-432); the captured graphs read the Adam
    moments / step counts from the same device buffers, so no re-capture is needed."""
    from .optim import load_merged_state_dict
    load_merged_state_dict(self.model, self.opt_flats + [self.opt_rest], sd)
    # the checkpoint's learning rate reaches EVERY optimizer of the step: a captured step keeps the rate of its torch Adam
    # in a device scalar, which load_merged_state_dict leaves alone (it only copies python-number hyper-parameters)
    lr = sd['param_groups'][0].get('lr')
    if lr is not None:
      self.set_lr(float(lr))

  def weights_changed(self):
    """Call after writing parameters behind the runner's back (model.load_state_dict on a runner that already exists, e.g.
    base/base_trainer.py:426-432 resuming into a live trainer): the captured step reads the GEMM weights from their bf16
    shadows, which only the optimizer -- or this call -- regenerates."""
    for o in self.opt_flats:
      o.flat.pack(force=True)

  def set_lr(self, lr):
    """One learning rate for every optimizer of the step (the reference has a single param group, train.py:100)."""
    for o in self.opt_flats:
      o.lr = lr
    if self.opt_rest is not None:
      if self._rest_lr is not None:
        self._rest_lr.fill_(float(lr))  # device scalar: the captured optimizer graph reads it at replay
      else:
        for g in self.opt_rest.param_groups:
          g['lr'] = lr


  def _fork_step(self):
    """One rank, fork mode: forward, then the backward stage by stage with the off-critical-path work on the side stream
    and the optimizer region by region behind it (module docstring).  Captured as ONE graph with parallel branches."""
    self._keep = []
    self._zero()
    e = self._forward()
    g = self._gather(e)
    self._regions = self._region_table()
    stages = self._stage_list(e, g)
    opt = self.opt_flat
    if self.fork & FORK_ADAM:
      opt._ensure_state()
      opt.sync_lr()
    for i, (fn, names) in enumerate(stages):
      fn()
      if self.fork & FORK_ADAM:
        self._adam_regions(names, last=i == len(stages) - 1)
    self._join()
    if not self.fork & FORK_ADAM:
      self._opt()

  # ---- the optimizer riding in the backward's GEMM launches (one rank) ------------------------------------------------
  def _arm_riders(self):
    """Build (once) the optimizer queues of the flat buffers and attach them to the encoders whose backward launches
    carry them.  Stages of the video side's queue: 0 = text heads (their backward runs first), 1 + k = the matrices and
    biases of encoder layer L-1-k (final after that layer's weight-gradient launch); LayerNorm parameters, embedding tables
    and the expert projections are only final when the backward ends and stay for the optimizer launch.  The native text
    tower's queue: stage k = its layer Lt-1-k; what its own backward leaves (embeddings, its bottom layer) rides in the
    video side's launches (chain)."""
    m = self.model
    opt_v = self.opt_flats[0]
    opt_t = self.opt_flats[1] if len(self.opt_flats) > 1 else None
    vb = m.vid_bert
    Lv = vb.config.num_hidden_layers
    built = False
    if opt_t is not None and not opt_t._queue_ok():
      tb = m.txt_bert
      Lt = tb.config.num_hidden_layers
      names = dict((id(p), n) for n, p in zip(opt_t.flat.names, opt_t.flat.params))
      opt_t.build_queue(lambda p: _layer_stage(names[id(p)], Lt, 0))
      built = True
    if built or not opt_v._queue_ok():
      names = dict((id(p), n) for n, p in zip(opt_v.flat.names, opt_v.flat.params))

      def stage_v(p):
        n = names[id(p)]
        if n.startswith('text_GU.') or n.startswith('moe_fc_txt.'):
          return 0
        return _layer_stage(n, Lv, 1)
      opt_v.build_queue(stage_v, chain=opt_t)
    for o in self.opt_flats:
      o.arm_queue(True)
    # stages of the queue that are final while layer l's backward runs: text heads + the layers above (video side), the
    # layers above (text tower)
    # (lab: MMT_RIDER_CAP = rider blocks at work per launch, MMT_RIDER_PASSES = passes a rider block makes, 0 = until the
    # host launch is in its tail; both travel in the upper half of rider_slot0 -> MmtEpilogue.rider_cap)
    cap = max(0, min(0xfff, int(os.environ.get('MMT_RIDER_CAP', '0'))))
    cap = (cap | (max(0, min(15, int(os.environ.get('MMT_RIDER_PASSES', '0')))) << 12)) << 16
    vb.set_rider(opt_v.queue_ptr(), [Lv - l for l in range(Lv)], cap)
    if opt_t is not None:
      tb = m.txt_bert
      Lt = tb.config.num_hidden_layers
      tb.set_rider(opt_t.queue_ptr(), [Lt - 1 - l for l in range(Lt)], cap)

  def _disarm_riders(self):
    self.model.vid_bert.set_rider(None)
    if len(self.opt_flats) > 1:
      self.model.txt_bert.set_rider(None)

  def _rider_step(self):
    """One rank: forward, the backward stage by stage (text heads first, then the encoder from the top layer down) with
    the optimizer queues riding in its GEMM launches, then the optimizer launches for what is left.  ONE serial chain of
    kernels, captured as one graph."""
    self._zero()
    e = self._forward()
    g = self._gather(e)
    self._regions = self._region_table()
    self._arm_riders()
    try:
      for fn, _ in self._stage_list(e, g, single_range=True):
        fn()
    finally:
      self._disarm_riders()
    self._opt()

  def _eager_step(self):
    if self._rider_on and not self._multi:
      return self._rider_step()
    if self._fork_on and not self._multi:
      return self._fork_step()
    self._zero()
    e = self._forward()
    g = self._gather(e)
    if self.staged:
      self._regions = self._region_table()
      handles = []
      for fn, names in self._stage_list(e, g):
        fn()
        handles += self._reduce_async(names)
      self._finish(handles)
    else:
      self.loss = self._loss_backward(e, g)
      self._sync_all()
    self._opt()

  # ---- capture -----------------------------------------------------------------------------------
  def _quiesce_watchdog(self):
    """RCCL's process group keeps every eager collective on a list that its watchdog thread polls (an event query per entry)
    until the collective has finished, and only then RETIRES the entry.  A poll that lands inside an open stream capture of
    THIS thread is legal in thread-local capture mode, but has been seen to end the process on ROCm (the watchdog thread
    dies with a c10::Error while the step graph is being captured).  Before a capture, therefore: finish the device's work,
    then WAIT until the watchdog has retired every entry -- observed through the process group's flight recorder
    (`_dump_nccl_trace(onlyActive=True)` lists exactly the collectives the watchdog still holds) -- so that it has nothing
    left to query while the capture is open.  Deterministic as long as the recorder is on (TORCH_NCCL_TRACE_BUFFER_SIZE > 0,
    the default); with the recorder off it falls back to waiting out a few of the watchdog's 100 ms poll periods, and says
    so once."""
    if not (dist.is_initialized() and dist.get_backend() == 'nccl'):
      return
    torch.cuda.synchronize()
    state = _pending_nccl_works()
    deadline = time.monotonic() + 10.0
    while state is not None and state > 0 and time.monotonic() < deadline:
      time.sleep(0.005)
      state = _pending_nccl_works()
    if state == 0:
      return  # every finished collective has left the watchdog's list
    global _QUIESCE_WARNED
    if not _QUIESCE_WARNED:
      _QUIESCE_WARNED = True
      import warnings
      warnings.warn('GraphedTrainStep: cannot observe the process group\'s pending collectives (%s); sleeping 0.35 s before '
                    'each capture instead' % ('flight recorder off' if state is None else 'entries still active after 10 s'))
    time.sleep(0.35)

  def _capture(self):
    torch.cuda.synchronize()
    self._zero()
    if not self._multi and (self._fork_on or self._rider_on):
      ga = torch.cuda.CUDAGraph()
      with torch.cuda.graph(ga, pool=self._pool, stream=self._stream, capture_error_mode=self._cap_mode):
        if self._rider_on:
          self._rider_step()  # ONE serial chain; the optimizer rides in the backward's GEMM launches
        else:
          self._fork_step()  # ONE graph whose branches are the main and the side stream
        self._upload_branch_end()
      self._graphs, self._e = (ga, None, None), None
      torch.cuda.synchronize()
      return
    ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    e = g = pool = None
    self._quiesce_watchdog()  # (the warm-up steps' collectives)
    if (self._multi or self.staged) and not (self._multi and self.capture_collectives):
      with torch.cuda.graph(ga, pool=self._pool, stream=self._stream, capture_error_mode=self._cap_mode):
        e = self._forward()
        self._upload_branch_end()
      pool = ga.pool()
      with torch.cuda.stream(self._stream):
        g = self._gather(e)
      self._quiesce_watchdog()  # (the eager all-gather just issued)
    if self._multi and self.capture_collectives:
      # EXPERIMENTAL (opt-in): the collectives are captured too, so a multi-rank step is ONE graph launch like the
      # single-rank one -- RCCL's stream joins the capture through the events torch.distributed records, the cross-stream
      # waits become graph edges.  Verified on a 1-rank RCCL group only (no multi-GPU box this round), hence not default.
      ga = torch.cuda.CUDAGraph()
      with torch.cuda.graph(ga, pool=self._pool, stream=self._stream, capture_error_mode=self._cap_mode):
        e = self._forward()
        g = self._gather(e)
        if self.staged:
          self._regions = self._region_table()
          handles = []
          for fn, names in self._stage_list(e, g):
            fn()
            handles += self._reduce_async(names)
          self._finish(handles)
        else:
          self.loss = self._loss_backward(e, g)
          self._sync_all()
        self._opt()
        self._upload_branch_end()
      self._graphs, self._e = (ga, None, None), e
      self._one_graph = True
      torch.cuda.synchronize()
      return
    if not self._multi and not self.staged:
      # nothing happens between forward and backward on one rank: one graph for both (one launch gap less per step)
      ga = torch.cuda.CUDAGraph()
      with torch.cuda.graph(ga, pool=self._pool, stream=self._stream, capture_error_mode=self._cap_mode):
        e = self._forward()
        g = self._gather(e)
        self.loss = self._loss_backward(e, g)
        self._opt()  # ... and the optimizer: the whole step is ONE graph launch
        self._upload_branch_end()
      self._graphs, self._e = (ga, None, None), e
      torch.cuda.synchronize()
      return
    elif self.staged:
      self._regions = self._region_table()
      gb = []
      for fn, names in self._stage_list(e, g):
        gs = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gs, pool=pool, stream=self._stream, capture_error_mode=self._cap_mode):
          fn()
        gb.append((gs, names))
    else:
      with torch.cuda.graph(gb, pool=pool, stream=self._stream, capture_error_mode=self._cap_mode):
        self.loss = self._loss_backward(e, g)
    if self.shard_opt and self.opt_rest is None:
      gc = None  # (the flat buffers were stepped shard by shard between the graphs: nothing left to capture)
    else:
      with torch.cuda.graph(gc, pool=pool, stream=self._stream, capture_error_mode=self._cap_mode):
        self._opt()
    self._graphs, self._e = (ga, gb, gc), e
    torch.cuda.synchronize()

  # ---- public ------------------------------------------------------------------------------------
  @property
  def input_slots(self):
    return len(self._statics)

  def inputs(self, slot=0):
    """The static input buffers of `slot` (what the captured step of that slot reads)."""
    return self._statics[slot]

  def load(self, minibatch, slot=0):
    """Copy a new minibatch (device tensors, same shapes) into the static input buffers of `slot`.  (Running this copy on
    a side stream under the previous step's optimizer graph was measured: the cross-stream event waits cost 70 us per
    step, more than the 14 us copy -- it stays on the compute stream.)"""
    static = self._statics[slot]
    if isinstance(minibatch, FlatMinibatch) and isinstance(static, FlatMinibatch) \
        and minibatch.flat.numel() == static.flat.numel():
      static.flat.copy_(minibatch.flat, non_blocking=True)
      for k in minibatch.ragged:
        static[k].copy_from(minibatch[k])
    else:
      _copy_tree(static, minibatch)

  def wait_upload(self, slot):
    """Host-side wait until the last `upload` into `slot` has left its pinned source buffer (before the loader refills
    that buffer)."""
    ev = self._slot_ev[slot] if self._slot_ev is not None else None
    if ev is not None and ev['ready'] is not None:
      ev['ready'].synchronize()

  def upload(self, minibatch, slot):
    """Start copying a HOST (pinned) FlatMinibatch straight into the input buffers of `slot` on a copy stream; the next
    `step(slot)` waits for it, and the upload itself waits until the previous `step(slot)` has been issued AND has run.
    With K >= 2 slots minibatch i + 1 crosses PCIe under step i and no device-to-device copy is left in the step."""
    static = self._statics[slot]
    if not isinstance(minibatch, FlatMinibatch) or not isinstance(static, FlatMinibatch):
      raise TypeError('upload needs FlatMinibatch inputs (one buffer per minibatch)')
    if self._slot_ev is None:
      self._slot_ev = [dict(ready=None, free=None) for _ in self._statics]
      self._copy_stream = torch.cuda.Stream()
    ev = self._slot_ev[slot]
    if ev['free'] is not None:
      # the last step that read this slot has finished (K - 1 steps back): waited for on the host as well, so that no
      # stream of the step ever waits for another stream
      if self.host_sync_uploads:
        ev['free'].synchronize()
      else:
        self._copy_stream.wait_event(ev['free'])
    else:
      self._copy_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(self._copy_stream):
      static.flat.copy_(minibatch.flat, non_blocking=True)
      for k in minibatch.ragged:
        static[k].copy_from(minibatch[k])
      if ev['ready'] is None:
        ev['ready'] = torch.cuda.Event()
      ev['ready'].record(self._copy_stream)
    ev['pending'] = True

  def prefetch(self, minibatch):
    """Start uploading a HOST (pinned) FlatMinibatch into a device staging buffer on a copy stream: the PCIe transfer
    of step i+1 runs under the kernels of step i.  `load_prefetched()` then moves it into the static inputs with a
    device-to-device copy on the compute stream."""
    if not isinstance(minibatch, FlatMinibatch) or not isinstance(self.static, FlatMinibatch):
      raise TypeError('prefetch needs FlatMinibatch inputs (one buffer per minibatch)')
    if self._staging is None:
      # two landing buffers: the upload of minibatch i+1 never waits for the device-side read of minibatch i
      self._staging = [dict(buf=FlatMinibatch(self.static, self.static.flat.device), ready=torch.cuda.Event(),
                            free=torch.cuda.Event()) for _ in range(2)]
      self._copy_stream = torch.cuda.Stream()
      for st in self._staging:
        st['free'].record(torch.cuda.current_stream())
      self._staged, self._uploads = [], 0
    st = self._staging[self._uploads % 2]
    self._uploads += 1
    self._copy_stream.wait_event(st['free'])  # the load_prefetched() two steps back has read this landing buffer
    with torch.cuda.stream(self._copy_stream):
      st['buf'].flat.copy_(minibatch.flat, non_blocking=True)
      for k in minibatch.ragged:
        st['buf'][k].copy_from(minibatch[k])
      st['ready'].record(self._copy_stream)
    self._staged.append(st)

  def load_prefetched(self):
    if not self._staged:
      raise RuntimeError('load_prefetched() without a prefetch()')
    st = self._staged.pop(0)
    cur = torch.cuda.current_stream()
    cur.wait_event(st['ready'])
    self.load(st['buf'])
    st['free'].record(cur)

  def eager_step(self):
    """One un-captured optimisation step on the runner's own stream (the stream the autograd graph's AccumulateGrad nodes
    are bound to: running the backward from another stream makes autograd insert cross-stream syncs and warn)."""
    cur = torch.cuda.current_stream()
    self._stream.wait_stream(cur)
    with torch.cuda.stream(self._stream):
      self._eager_step()
    cur.wait_stream(self._stream)
    return self.loss

  def step(self, slot=0, live_rows=None):
    """Runs one optimisation step on the static inputs (of `slot`); returns the (device) loss tensor.  live_rows: the
    minibatch's packed token rows as the loader counted them (see the constructor)."""
    if not self.use_graphs:
      if live_rows is not None and hasattr(self.model, 'live_rows_hint'):
        self.model.live_rows_hint = int(live_rows)
      return self.eager_step()
    if live_rows is not None and not self._multi:
      # (one rank only: a re-capture of the multi-rank step issues collectives -- the eager all-gather between its graphs --
      # and only the ranks whose minibatch changed bucket would enter them; under data parallelism the tiles stay the ones of
      # the constructor's count)
      sig = self._tile_signature(live_rows)
      if sig is not None and sig != self._sig:
        self._switch_tiles(sig, live_rows)
    for o in self.opt_flats:
      o.sync_lr()  # learning-rate schedule: the captured optimizer graph reads the rate from the device
    if self._caps is not None:
      self._graphs, self._e, self.loss, self._one_graph = self._caps[slot]
    elif slot:
      raise ValueError('step(slot=%d): the runner was built with one input slot' % slot)
    ev = self._slot_ev[slot] if self._slot_ev is not None else None
    if ev is not None and ev.get('pending'):
      # The upload of this slot was enqueued a whole step ago.  Waiting for it on the HOST (normally already complete: a
      # query) keeps the compute stream free of cross-stream dependencies: a hipStreamWaitEvent in front of the graph
      # launch costs ~0.1 ms of queue plumbing per step on this runtime, whatever the bytes (tools/feed_lab.py).
      if self.host_sync_uploads:
        ev['ready'].synchronize()
      else:
        torch.cuda.current_stream().wait_event(ev['ready'])
      ev['pending'] = False
    try:
      return self._replay()
    finally:
      if self._host_feed is not None:
        if self._done_ev is None:
          self._done_ev = [None] * len(self._statics)
        if self._done_ev[slot] is None:
          self._done_ev[slot] = torch.cuda.Event()
        self._done_ev[slot].record(torch.cuda.current_stream())
      if ev is not None:
        if ev['free'] is None:
          ev['free'] = torch.cuda.Event()
        ev['free'].record(torch.cuda.current_stream())

  def _replay(self):
    ga, gb, gc = self._graphs
    ga.replay()
    if self._multi and not self._one_graph:
      self._gather(self._e)
    if gb is None:
      pass  # forward + backward were captured as one graph
    elif self.staged:
      handles = []
      for gs, names in gb:
        gs.replay()
        handles += self._reduce_async(names)
      ev = self._exposed_pair()
      if ev:
        ev[0].record()
      self._finish(handles)
      if ev:
        ev[1].record()
    else:
      gb.replay()
      ev = self._exposed_pair()
      if ev:
        ev[0].record()
      self._sync_all()
      if ev:
        ev[1].record()
    if gc is not None:
      gc.replay()
    return self.loss

  measure_exposed = False
  host_sync_uploads = True

  def _exposed_pair(self):
    if not (self.measure_exposed and self._multi):
      return None
    pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    self._exposed.append(pair)
    if len(self._exposed) > 512:
      self._exposed.pop(0)
    return pair

  def exposed_collective_ms(self):
    """Mean time per step the compute stream spent between the end of the last backward stage and the point where every
    gradient reduction has landed (HIP events on the compute stream around the final waits; `measure_exposed = True`
    before the steps, synchronize before reading): the part of the gradient exchange the backward did NOT hide."""
    ms = [a.elapsed_time(b) for a, b in self._exposed if a.query() and b.query()]
    return sum(ms) / len(ms) if ms else None
