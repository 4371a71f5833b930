"""Drop-in for the reference's `model/bert.py` BertModel running on the MI355X engine.

Same constructor (`BertModel(config_namespace)`), same `forward(input_ids, attention_mask,
token_type_ids, position_ids, features)` signature and tuple return `(sequence_output,
pooled_output)` (model/bert.py:371-414), same parameter names/shapes/initialisation
(model/bert.py:74-86,116-121,178-183,211-215,226-231,298-301,361-369) so reference checkpoints load.
The sub-modules below are parameter containers only: the arithmetic of bert.py:87-292 runs in
libmmt_hip.so (one C call forward, one backward); there is no torch fallback.

Numerics: bf16 MFMA GEMM operands with fp32 accumulation; residual stream, LayerNorm statistics,
softmax and gradients wrt hidden states in fp32 (DESIGN.md states the resulting tolerances).
"""
import ctypes
import math

import torch
from torch import nn

from . import _lib, ops
from ._lib import MmtBertBatch, MmtBertLayer, MmtBertModel, check
from .flat import FlatParams


class BertEmbeddings(nn.Module):
  """Parameter container for model/bert.py:74-86."""

  def __init__(self, config):
    super().__init__()
    self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
    self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
    self.layer_norm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
    self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertSelfAttention(nn.Module):

  def __init__(self, config):
    super().__init__()
    if config.hidden_size % config.num_attention_heads != 0:
      raise ValueError('The hidden size (%d) is not a multiple of the number of attention heads (%d)' %
                       (config.hidden_size, config.num_attention_heads))
    self.num_attention_heads = config.num_attention_heads
    self.attention_head_size = config.hidden_size // config.num_attention_heads
    self.query = nn.Linear(config.hidden_size, config.hidden_size)
    self.key = nn.Linear(config.hidden_size, config.hidden_size)
    self.value = nn.Linear(config.hidden_size, config.hidden_size)
    self.dropout = nn.Dropout(config.attention_probs_dropout_prob)


class BertSelfOutput(nn.Module):

  def __init__(self, config):
    super().__init__()
    self.dense = nn.Linear(config.hidden_size, config.hidden_size)
    self.layer_norm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
    self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertAttention(nn.Module):

  def __init__(self, config):
    super().__init__()
    self.self = BertSelfAttention(config)
    self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):

  def __init__(self, config):
    super().__init__()
    if config.hidden_act != 'gelu':
      raise NotImplementedError('native video-BERT implements hidden_act="gelu" (erf form) only')
    self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class BertOutput(nn.Module):

  def __init__(self, config):
    super().__init__()
    self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
    self.layer_norm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
    self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertLayer(nn.Module):

  def __init__(self, config):
    super().__init__()
    self.attention = BertAttention(config)
    self.intermediate = BertIntermediate(config)
    self.output = BertOutput(config)


class BertEncoder(nn.Module):

  def __init__(self, config):
    super().__init__()
    self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])


class BertPooler(nn.Module):
  """model/bert.py:295-309; evaluated with stock torch ops on request only (CENet ignores it)."""

  def __init__(self, config):
    super().__init__()
    self.dense = nn.Linear(config.hidden_size, config.hidden_size)
    self.activation = nn.Tanh()

  def forward(self, hidden_states):
    return self.activation(self.dense(hidden_states[:, 0]))


class EngineBatch:
  """Device-side description of one minibatch for the engine (token rows may be packed)."""

  def __init__(self, features, type_ids, pos_ids, mask_bias, rows, batch, seq, cu_seqlens=None,
               row_index=None, n_rows_dev=None, out_rows=None, n_out_per_sample=0):
    self.features, self.type_ids, self.pos_ids, self.mask_bias = features, type_ids, pos_ids, mask_bias
    self.rows, self.batch, self.seq = rows, batch, seq
    self.cu_seqlens, self.row_index, self.n_rows_dev = cu_seqlens, row_index, n_rows_dev
    # optional: the only rows of sequence_output the caller reads (int32 [batch * n_out_per_sample]); the last layer
    # is then evaluated on those rows only and the other rows of the returned tensor are undefined
    self.out_rows, self.n_out_per_sample = out_rows, n_out_per_sample
    self.save = False
    # backward only (MmtBertBatch.fork / side_stream): a second stream (torch.cuda.Stream) for the launches that are
    # off the critical path of the step, and the _lib.FORK_* bits that say which
    self.side_stream, self.fork = None, 0


class _BertFn(torch.autograd.Function):
  """sequence_output = video_BERT(features); parameters are passed so autograd routes their grads."""

  @staticmethod
  def forward(ctx, model, batch, features, *params):
    out = model._engine_forward(batch, features, save=batch.save)
    ctx.model, ctx.batch, ctx.generation = model, batch, model._generation
    ctx.training = model.training
    return out

  @staticmethod
  def backward(ctx, dout):
    model = ctx.model
    if model._generation != ctx.generation:
      raise RuntimeError('mmt_amd.BertModel: saved activations were overwritten by a later forward '
                         '(one pending backward per module)')
    dfeat, grads = model._engine_backward(ctx.batch, dout, ctx.training)
    return (None, None, dfeat) + tuple(grads)


class BertModel(nn.Module):
  r"""Multi-modal video BERT (no word embeddings), MI355X-native.  See module docstring."""

  def __init__(self, config):
    super().__init__()
    self.config = config
    self.embeddings = BertEmbeddings(config)
    self.encoder = BertEncoder(config)
    self.pooler = BertPooler(config)
    self.apply(self._init_weights)
    self.compute_pooler = True
    self._flat = None
    self._owns_flat = True
    self._ws = {}
    self._generation = 0
    self._seed_dev = None
    self._structs = {}

  def _init_weights(self, module):
    """model/bert.py:361-369."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
      module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
    elif isinstance(module, nn.LayerNorm):
      module.bias.data.zero_()
      module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
      module.bias.data.zero_()

  # ---- flat parameter layout ---------------------------------------------------------------------
  def engine_named_params(self, prefix=''):
    e = self.embeddings
    out = [('embeddings.position_embeddings.weight', e.position_embeddings.weight),
           ('embeddings.token_type_embeddings.weight', e.token_type_embeddings.weight),
           ('embeddings.layer_norm.weight', e.layer_norm.weight), ('embeddings.layer_norm.bias', e.layer_norm.bias)]
    for i, L in enumerate(self.encoder.layer):
      a, p = L.attention, 'encoder.layer.%d.' % i
      out += [(p + 'attention.self.query.weight', a.self.query.weight),
              (p + 'attention.self.key.weight', a.self.key.weight),
              (p + 'attention.self.value.weight', a.self.value.weight),
              (p + 'attention.self.query.bias', a.self.query.bias),
              (p + 'attention.self.key.bias', a.self.key.bias),
              (p + 'attention.self.value.bias', a.self.value.bias),
              (p + 'attention.output.dense.weight', a.output.dense.weight),
              (p + 'attention.output.dense.bias', a.output.dense.bias),
              (p + 'attention.output.layer_norm.weight', a.output.layer_norm.weight),
              (p + 'attention.output.layer_norm.bias', a.output.layer_norm.bias),
              (p + 'intermediate.dense.weight', L.intermediate.dense.weight),
              (p + 'intermediate.dense.bias', L.intermediate.dense.bias),
              (p + 'output.dense.weight', L.output.dense.weight), (p + 'output.dense.bias', L.output.dense.bias),
              (p + 'output.layer_norm.weight', L.output.layer_norm.weight),
              (p + 'output.layer_norm.bias', L.output.layer_norm.bias)]
    return [(prefix + n, q) for n, q in out]

  def trainable_engine_params(self):
    return [p for _, p in self.engine_named_params()]

  def flat_named_params(self):
    """Everything stored in this module's own flat buffer (subclasses add parameters the engine does not touch)."""
    return self.engine_named_params()

  def register_shadows(self, flat):
    d, i = self.config.hidden_size, self.config.intermediate_size
    for l, L in enumerate(self.encoder.layer):
      a = L.attention
      flat.add_shadow(('bert', id(self), l, 'wqkv'), [a.self.query.weight, a.self.key.weight, a.self.value.weight],
                      3 * d, d, transpose=True)
      flat.add_shadow(('bert', id(self), l, 'wo'), [a.output.dense.weight], d, d, transpose=True)
      flat.add_shadow(('bert', id(self), l, 'w1'), [L.intermediate.dense.weight], i, d, transpose=True)
      flat.add_shadow(('bert', id(self), l, 'w2'), [L.output.dense.weight], d, i, transpose=True)

  def attach_flat(self, flat):
    """Use an externally owned FlatParams (CENet shares one buffer for the whole video side)."""
    self._flat, self._owns_flat = flat, False
    self._structs = {}

  def build_flat(self):
    """The flat parameter buffer this module owns (created on first use)."""
    if self._flat is None:
      self._flat = FlatParams(self.flat_named_params())
      self.register_shadows(self._flat)
    return self._flat

  def _ensure_ready(self, device):
    self.build_flat()
    if self._owns_flat:
      if self._flat.ensure(device):
        self._structs = {}
      self._flat.pack()
      if torch.is_grad_enabled():
        self._flat.select_grad_buffer()
    if self._seed_dev is None or self._seed_dev.device != torch.device(device):
      seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
      # data-parallel replicas draw INDEPENDENT dropout masks (the reference's replicas each run their own RNG stream,
      # trainer/trainer.py:134): the masks are keyed on rank-local coordinates, so the rank is folded into the seed --
      # identical torch seeds on every rank (bench.py, a trainer calling torch.manual_seed) no longer mean identical masks
      import torch.distributed as dist
      if dist.is_available() and dist.is_initialized():
        seed = (seed + 0x3C6EF372 * (dist.get_rank() + 1)) % (2 ** 31 - 1)
      self._seed_dev = torch.tensor([seed], dtype=torch.int32, device=device)

  def _struct(self, grad_buf):
    """MmtBertModel (+ layer array) pointing into flat master / shadows / `grad_buf`."""
    f = self._flat
    key = (f.master.data_ptr(), grad_buf.data_ptr())
    hit = self._structs.get(key)
    if hit is not None:
      return hit
    cfg = self.config
    layers = (MmtBertLayer * cfg.num_hidden_layers)()
    for l, L in enumerate(self.encoder.layer):
      a, s = L.attention, layers[l]
      for name in ('wqkv', 'wo', 'w1', 'w2'):
        w, wt = f.shadow(('bert', id(self), l, name))
        setattr(s, name, w.data_ptr())
        setattr(s, name + '_t', wt.data_ptr())
      pairs = dict(bqkv=a.self.query.bias, bo=a.output.dense.bias, ln1_g=a.output.layer_norm.weight,
                   ln1_b=a.output.layer_norm.bias, b1=L.intermediate.dense.bias, b2=L.output.dense.bias,
                   ln2_g=L.output.layer_norm.weight, ln2_b=L.output.layer_norm.bias)
      for name, p in pairs.items():
        setattr(s, name, f.ptr(p))
        setattr(s, 'g_' + name, f.ptr(p, grad_buf))
      for name, p in dict(wqkv=a.self.query.weight, wo=a.output.dense.weight, w1=L.intermediate.dense.weight,
                          w2=L.output.dense.weight).items():
        setattr(s, 'g_' + name, f.ptr(p, grad_buf))
    m = MmtBertModel()
    m.hidden, m.layers, m.heads = cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads
    m.inter, m.max_pos, m.type_vocab = cfg.intermediate_size, cfg.max_position_embeddings, cfg.type_vocab_size
    m.ln_eps, m.p_hidden, m.p_attn = cfg.layer_norm_eps, cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob
    e = self.embeddings
    for name, p in dict(pos_emb=e.position_embeddings.weight, type_emb=e.token_type_embeddings.weight,
                        emb_ln_g=e.layer_norm.weight, emb_ln_b=e.layer_norm.bias).items():
      setattr(m, name, f.ptr(p))
      setattr(m, 'g_' + name, f.ptr(p, grad_buf))
    m.layer = layers
    self._structs[key] = (m, layers)
    return m, layers

  # ---- engine calls ------------------------------------------------------------------------------
  def _batch_struct(self, batch, rows_alloc):
    b = MmtBertBatch()
    b.features = batch.features.data_ptr()
    b.type_ids = batch.type_ids.data_ptr()
    b.pos_ids = batch.pos_ids.data_ptr() if batch.pos_ids is not None else None
    b.mask_bias = batch.mask_bias.data_ptr()
    b.cu_seqlens = batch.cu_seqlens.data_ptr() if batch.cu_seqlens is not None else None
    b.row_index = batch.row_index.data_ptr() if batch.row_index is not None else None
    b.n_rows_dev = batch.n_rows_dev.data_ptr() if batch.n_rows_dev is not None else None
    b.seed_dev = self._seed_dev.data_ptr()
    b.rows, b.rows_alloc, b.batch, b.seq = batch.rows, rows_alloc, batch.batch, batch.seq
    if batch.out_rows is not None and batch.n_out_per_sample > 0:
      b.out_rows, b.n_out_per_sample = batch.out_rows.data_ptr(), batch.n_out_per_sample
    if batch.side_stream is not None and batch.fork:
      b.side_stream, b.fork = batch.side_stream.cuda_stream, int(batch.fork)
    r = self._rider
    if r is not None:  # the optimizer queue this module's backward GEMMs carry (set_rider)
      b.rider, b.rider_limits, b.rider_slot0 = r['ptr'], ctypes.addressof(r['limits']), r['slot0']
    b.live_rows_hint = int(getattr(batch, 'live_rows_hint', 0) or 0)
    return b

  _rider = None

  def set_rider(self, queue_ptr=None, limits=None, slot0=0):
    """Attach (queue_ptr = device address of an MmtAdamQueue) or detach (None) the optimizer queue the GEMM launches of
    this module's BACKWARD carry; limits[l] = queue entries whose gradients are final when layer l's backward starts
    (include/mmt_hip.h: MmtBertBatch.rider).  Only a caller that finishes the queue after every backward
    (FlatAdam.step with an armed queue: train_step.GraphedTrainStep) may attach one -- riders UPDATE WEIGHTS."""
    if queue_ptr is None:
      self._rider = None
      return
    n = self.config.num_hidden_layers
    if len(limits) != n:
      raise ValueError('set_rider: one limit per layer')
    self._rider = dict(ptr=int(queue_ptr), limits=(ctypes.c_int32 * n)(*[int(x) for x in limits]), slot0=int(slot0))

  def _workspace(self, rows_alloc, save, model_struct):
    key = (rows_alloc, bool(save))
    ws = self._ws.get(key)
    if ws is None:
      nbytes = _lib.lib().mmt_bert_workspace_bytes(ctypes.byref(model_struct), rows_alloc)
      if nbytes <= 0:
        raise RuntimeError('mmt_bert_workspace_bytes failed (%d)' % nbytes)
      ws = torch.zeros(nbytes, dtype=torch.uint8, device=self._flat.master.device)
      self._ws[key] = ws
    return ws

  def compact_output(self, batch, rows_alloc):
    """True iff the engine returns only the batch.out_rows rows, compacted (see MmtBertBatch.out_rows)."""
    if batch.out_rows is None or batch.n_out_per_sample <= 0:
      return False
    return batch.batch * batch.n_out_per_sample <= _lib.lib().mmt_bert_tail_capacity(rows_alloc)

  def _engine_forward(self, batch, features, save):
    rows_alloc = features.shape[0]
    if features.dtype != torch.float32 or not features.is_contiguous() or rows_alloc % ops.ROW_ALIGN:
      raise RuntimeError('engine features must be contiguous fp32 [rows padded to %d, hidden]' % ops.ROW_ALIGN)
    batch.features = features
    grad_buf = self._flat.current_grad()
    m, _ = self._struct(grad_buf)
    if save:
      self._generation += 1
      if self.training and not getattr(self, '_seed_bumped', False):
        self._seed_dev.add_(1)
    self._seed_bumped = False  # (CENet's token-plan kernel increments the seed itself when it runs in front of us)
    ws = self._workspace(rows_alloc, save, m)
    out = torch.empty(rows_alloc, self.config.hidden_size, device=features.device, dtype=torch.float32)
    b = self._batch_struct(batch, rows_alloc)
    check(_lib.lib().mmt_bert_forward(ctypes.byref(m), ctypes.byref(b), ws.data_ptr(), out.data_ptr(),
                                      int(self.training), ops._stream()), 'mmt_bert_forward')
    return out

  def _engine_backward(self, batch, dout, training):
    run = self.backward_ranges(batch, dout, training)
    run(self.config.num_hidden_layers - 1, 0)
    grad_buf = self._flat.current_grad()
    grads = [self._flat.view(p, grad_buf) if p.requires_grad else None for p in self.trainable_engine_params()]
    return run.dfeat, grads

  def backward_ranges(self, batch, dout, training):
    """-> run(l_hi, l_lo): the engine backward of layers l_hi .. l_lo (descending; embeddings with layer 0), to be
    called range by range from the top layer down (mmt_bert_backward_range); run.dfeat holds the gradient wrt the
    engine's input features after the range that ends at layer 0.  Parameter gradients land in the flat buffer."""
    rows_alloc = batch.features.shape[0]
    grad_buf = self._flat.current_grad()
    m, _ = self._struct(grad_buf)
    ws = self._workspace(rows_alloc, True, m)
    if self.compact_output(batch, rows_alloc):
      dlast = dout.contiguous()  # compact read-out gradient in the first rows; produced by our own read-out backward,
                                 # so the engine may use the rest of it (and later all of it) as scratch
    else:
      dlast = dout.contiguous().clone()  # the engine uses it as scratch
    dfeat = torch.empty_like(dlast)
    L = _lib.lib()

    def run(l_hi, l_lo, embed=True):
      """embed=False with l_lo == 0: stop before the embedding stage; run(-1, -1) then runs that stage alone."""
      b = self._batch_struct(batch, rows_alloc)  # (per call: the caller may change batch.fork between ranges)
      if not embed:
        b.fork |= _lib.RANGE_LAYERS_ONLY
      check(L.mmt_bert_backward_range(ctypes.byref(m), ctypes.byref(b), ws.data_ptr(), dlast.data_ptr(),
                                      dfeat.data_ptr(), int(training), int(l_hi), int(l_lo), ops._stream()),
            'mmt_bert_backward_range')

    run.dfeat, run.keep = dfeat, (dlast, ws, m)
    return run

  def run_engine(self, batch, features):
    """features: fp32 [rows_alloc, hidden] (may require grad) -> sequence_output rows [rows_alloc, hidden]."""
    self._ensure_ready(features.device)
    params = self.trainable_engine_params()
    batch.save = torch.is_grad_enabled() and (features.requires_grad or any(p.requires_grad for p in params))
    return _BertFn.apply(self, batch, features, *params)

  # ---- reference-compatible forward --------------------------------------------------------------
  def forward(self, input_ids, attention_mask=None, token_type_ids=None, position_ids=None, features=None):
    """model/bert.py:371-414.  `input_ids` only provides the (B, S) shape, as in the reference."""
    if features is None:
      raise ValueError('features are required (the video BERT has no word embeddings)')
    if not features.is_cuda:
      raise RuntimeError('mmt_amd.BertModel runs on the GPU only (no CPU fallback)')
    bsz, seq, d = features.shape
    dev = features.device
    rows = bsz * seq
    R = ops.pad_rows(rows)
    if attention_mask is None:
      attention_mask = torch.ones(bsz, seq, device=dev)
    if token_type_ids is None:
      token_type_ids = torch.zeros(bsz, seq, dtype=torch.long, device=dev)

    # the reference's nn.Embedding lookups raise IndexError on ids outside their tables (bert.py:96-99); the fused
    # embedding kernel would read out of bounds, so this (host-synchronising) check guards the standalone entry point --
    # CENet's own path builds its ids on the device and clamps them (assemble.hip)
    for name, ids, size in (('token_type_ids', token_type_ids, self.config.type_vocab_size),
                            ('position_ids', position_ids, self.config.max_position_embeddings)):
      if ids is not None and ids.numel() and not (0 <= int(ids.min()) and int(ids.max()) < size):
        raise IndexError('%s out of range [0, %d)' % (name, size))

    def rows_i32(x):
      buf = torch.zeros(R, dtype=torch.int32, device=dev)
      buf[:rows] = x.reshape(-1).to(device=dev, dtype=torch.int32)
      return buf

    feat = torch.zeros(R, d, device=dev, dtype=torch.float32)
    feat[:rows] = features.reshape(rows, d).float()
    if features.requires_grad:
      feat = _PadRows.apply(features.reshape(rows, d).float(), R)
    mask_bias = torch.zeros(R, device=dev, dtype=torch.float32)
    mask_bias[:rows] = (1.0 - attention_mask.reshape(-1).to(device=dev, dtype=torch.float32)) * -10000.0
    batch = EngineBatch(None, rows_i32(token_type_ids), rows_i32(position_ids) if position_ids is not None else None,
                        mask_bias, rows, bsz, seq)
    seq_out = self.run_engine(batch, feat)[:rows].view(bsz, seq, d)
    if not self.compute_pooler:
      return (seq_out, None)
    return (seq_out, self.pooler(seq_out))


class _PadRows(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, R):
    ctx.rows = x.shape[0]
    out = torch.zeros(R, x.shape[1], device=x.device, dtype=x.dtype)
    out[:ctx.rows] = x
    return out

  @staticmethod
  def backward(ctx, g):
    return g[:ctx.rows], None
