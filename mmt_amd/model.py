"""Drop-in for the reference's `model/model.py` (CENet + sharded_cross_view_inner_product) on MI355X.

`CENet(**arch_args, expert_dims=..., tokenizer=...)` keeps the reference constructor
(model/model.py:48-73), `forward(token_ids, features, features_t, features_ind, features_avgpool,
features_maxpool, query_masks, out, device, debug)` (model/model.py:312-322), the returned dicts
(model/model.py:633-661) and every parameter name, so `train.py` / `trainer/trainer.py` drive it
unchanged and released checkpoints load.

What runs where:
  * video side -- ReduceDim per expert, token assembly, video BERT, expert read-out + L2 norm
    (model/model.py:426-437, 485-587, 621-625): libmmt_hip.so, forward and backward;
  * similarity (model/model.py:789-837) and the losses: libmmt_hip.so;
  * text heads (GatedEmbeddingUnit per expert, text MoE weights, model/model.py:229-283, 683-750): libmmt_hip.so
    (texthead.hip / texthead2.hip), forward and backward, fp32 on the exact-fp32 matrix cores;
  * text tower: `mmt_amd.text_bert.TextBertModel` on the same engine (HF parameter names; txt_bert='native' or a
    pretrained HF BertModel moved over by TextBertModel.from_hf); a foreign `txt_bert` module is accepted and then runs as
    stock PyTorch-ROCm outside the native path.
Only the configuration every published config uses is implemented natively (vid_cont='bert',
vid_inp='both', out_tok='mxp', pos_enc='tint'|'none', vid_wgh='none', keep_missing_modalities=True);
anything else raises NotImplementedError instead of silently taking a slow path.
"""
import collections
import ctypes
import os
import re
import types

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, ops
from ._lib import MmtExpertIO, MmtTextHeads, MmtTextHeadsOpts, MmtVideoFront, MmtVideoSrc, check
from .bert import BertModel, EngineBatch
from .feature_store import RaggedFeatures
from .flat import FlatParams


def _round_up(x, m):
  return (x + m - 1) // m * m


class ReduceDim(nn.Module):
  """model/model.py:717-726 (parameter container; also usable as a plain torch module for text)."""

  def __init__(self, input_dimension, output_dimension):
    super().__init__()
    self.fc = nn.Linear(input_dimension, output_dimension)

  def forward(self, x):
    return F.normalize(self.fc(x), dim=-1)


class ContextGating(nn.Module):
  """model/model.py:736-750."""

  def __init__(self, dimension, add_batch_norm=True):
    super().__init__()
    self.fc = nn.Linear(dimension, dimension)
    self.add_batch_norm = add_batch_norm
    self.batch_norm = nn.BatchNorm1d(dimension)

  def forward(self, x):
    x1 = self.fc(x)
    if self.add_batch_norm:
      x1 = self.batch_norm(x1)
    return F.glu(torch.cat((x, x1), 1), 1)


class GatedEmbeddingUnit(nn.Module):
  """model/model.py:683-702."""

  def __init__(self, input_dimension, output_dimension, use_bn, normalize):
    super().__init__()
    self.fc = nn.Linear(input_dimension, output_dimension)
    self.cg = ContextGating(output_dimension, add_batch_norm=use_bn)
    self.normalize = normalize

  def forward(self, x):
    x = self.cg(self.fc(x))
    if self.normalize:
      x = F.normalize(x, dim=-1)
    return x


# ------------------------------------------------------------------------------------------------
# native autograd functions
# ------------------------------------------------------------------------------------------------
class _VideoTokensFn(torch.autograd.Function):
  """features[row] = F.normalize(ReduceDim.fc(expert feature)) scattered into (packed) token rows."""

  @staticmethod
  def forward(ctx, net, plan, *params):
    ctx.net, ctx.plan, ctx.generation = net, plan, plan.generation
    return net._video_tokens_forward(plan)

  @staticmethod
  def backward(ctx, dfeat):
    if ctx.plan.generation != ctx.generation:
      raise RuntimeError('mmt_amd.CENet: video-token buffers were overwritten by a later forward')
    return (None, None) + tuple(ctx.net._video_tokens_backward(ctx.plan, dfeat.contiguous()))


class _ReadoutFn(torch.autograd.Function):
  """vid_embds[b, m] = F.normalize(sequence_output[agg_row[b, m]])  (model.py:583-587, 621-625)."""

  @staticmethod
  def forward(ctx, last, agg_row, bm, compact=False, stash=None):
    """stash (dict, optional): receives the inverse norms, for callers that fuse this backward into theirs
    (train_step.GraphedTrainStep: loss + similarity backward + read-out backward in one launch)."""
    ctx.compact = compact
    d = last.shape[1]
    out = torch.empty(bm, d, device=last.device, dtype=torch.float32)
    inv = torch.empty(bm, device=last.device, dtype=torch.float32)
    check(_lib.lib().mmt_readout_fwd(ops._p(last), ops._p(agg_row), bm, d, ops._p(out), ops._p(inv), ops._stream()),
          'mmt_readout_fwd')
    ctx.save_for_backward(out, inv, agg_row)
    ctx.shape = last.shape
    if stash is not None:
      stash.update(readout_inv=inv, readout_rows=None if compact else agg_row, readout_compact=compact)
    return out

  @staticmethod
  def backward(ctx, dout):
    out, inv, agg_row = ctx.saved_tensors
    # compact engine output: only the first B*M rows exist, no zero fill of the other token rows is needed
    dlast = (torch.empty if ctx.compact else torch.zeros)(ctx.shape, device=out.device, dtype=torch.float32)
    check(_lib.lib().mmt_readout_bwd(ops._p(out), ops._p(inv), ops._p(dout.contiguous()), ops._p(agg_row),
                                     out.shape[0], out.shape[1], ops._p(dlast), ops._stream()), 'mmt_readout_bwd')
    return dlast, None, None, None, None


class _SimsFn(torch.autograd.Function):
  """sims[t, v] of model/model.py:789-837 for txt [NT,M,d], vid [NV,M,d], tw [NT,M], vw [NV,M]."""

  @staticmethod
  def forward(ctx, txt, vid, tw, vw):
    nt, m, d = txt.shape
    nv = vid.shape[0]
    txt, vid, tw, vw = (x.detach().contiguous().float() for x in (txt, vid, tw, vw))
    sims = torch.empty(nt, nv, device=txt.device, dtype=torch.float32)
    dots = torch.empty(nt, nv, m, device=txt.device, dtype=torch.float32)
    check(_lib.lib().mmt_sims_fwd(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), nt, nv, m, d, ops._p(sims),
                                  ops._p(dots), ops._stream()), 'mmt_sims_fwd')
    ctx.save_for_backward(txt, vid, tw, vw, dots)
    return sims

  @staticmethod
  def backward(ctx, dsims):
    txt, vid, tw, vw, dots = ctx.saved_tensors
    nt, m, d = txt.shape
    nv = vid.shape[0]
    dtxt, dvid, dtw, dvw = (torch.empty_like(x) for x in (txt, vid, tw, vw))
    check(_lib.lib().mmt_sims_bwd(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), ops._p(dots),
                                  ops._p(dsims.contiguous().float()), nt, nv, m, d, ops._p(dtxt), ops._p(dvid),
                                  ops._p(dtw), ops._p(dvw), ops._stream()), 'mmt_sims_bwd')
    return dtxt, dvid, dtw, dvw


class _MoeDropoutFn(torch.autograd.Function):
  """nn.Dropout in front of the text MoE logits (model/model.py:274) on the engine's counter-based RNG: with it a
  captured training step contains no framework RNG (a philox dropout makes every graph replay refill the generator's
  seed / offset tensors first: two extra launches per step)."""
  SITE_KEY = 0x7e57d0a1

  @staticmethod
  def forward(ctx, x, p, seed_dev):
    x = x.contiguous().float()
    y = torch.empty_like(x)
    ctx.key = torch.empty(1, dtype=torch.int32, device=x.device)
    ctx.thr, ctx.scale = ops.dropout_params(p)
    check(_lib.lib().mmt_dropout_f32(ops._p(x), ops._p(y), x.numel(), _MoeDropoutFn.SITE_KEY, ctx.thr, ctx.scale,
                                     ops._p(seed_dev), ops._p(ctx.key), None, ops._stream()), 'mmt_dropout_f32')
    return y

  @staticmethod
  def backward(ctx, g):
    g = g.contiguous().float()
    gx = torch.empty_like(g)
    check(_lib.lib().mmt_dropout_f32(ops._p(g), ops._p(gx), g.numel(), _MoeDropoutFn.SITE_KEY, ctx.thr, ctx.scale, None,
                                     None, ops._p(ctx.key), ops._stream()), 'mmt_dropout_f32')
    return gx, None, None


class _TextHeadsFn(torch.autograd.Function):
  """text (B*C, K) -> text_embds (B, M, C, d), text_weights (B, C, M): the per-expert GatedEmbeddingUnits
  (model.py:413-417, 683-750) and the text MoE softmax (model.py:262-283, 610-618) in one native pass."""

  @staticmethod
  def forward(ctx, net, text, text_moe, caps, moe_drop_p, *params):
    """moe_drop_p > 0 (text_moe None): moe_txt_dropout (model.py:274) is applied on the fly inside the kernels."""
    tm = None if text_moe is None else text_moe.detach().contiguous().float()
    out = net._text_heads_forward(text.detach().contiguous().float(), tm, caps, moe_drop_p)
    ctx.net, ctx.caps, ctx.generation, ctx.training = net, caps, net._th_generation, net.training
    ctx.need_dtext = text.requires_grad
    ctx.need_dmoe = text_moe is not None and text_moe.requires_grad
    net._th_needs_input_grad = ctx.need_dtext or ctx.need_dmoe  # (train_step: may the backward bypass autograd?)
    return out

  @staticmethod
  def backward(ctx, de, dtw):
    net = ctx.net
    if net._th_generation != ctx.generation:
      raise RuntimeError('mmt_amd.CENet: text-head buffers were overwritten by a later forward')
    dtext, dmoe, grads = net._text_heads_backward(ctx.caps, de, dtw, ctx.need_dtext, ctx.need_dmoe, ctx.training)
    return (None, dtext, dmoe, None, None) + tuple(grads)


def cross_view_similarity(vid_embds, text_embds, vid_weights, text_weights, merge='avg'):
  """Tensor form of sharded_cross_view_inner_product: vid (B,M,d), text (B,M,C,d), vw (B,M), tw (B,C,M)."""
  b, m, d = vid_embds.shape
  c = text_embds.shape[2]
  dev = vid_embds.device
  on_cpu = not vid_embds.is_cuda
  if on_cpu:  # eval path of the trainer hands CPU tensors (trainer.py:368,396): round-trip through the GPU
    if not torch.cuda.is_available():
      raise RuntimeError('mmt_amd similarity needs a GPU (no CPU fallback)')
    vid_embds, text_embds, vid_weights, text_weights = (x.cuda() for x in (vid_embds, text_embds, vid_weights,
                                                                           text_weights))
  txt = text_embds.permute(0, 2, 1, 3).reshape(b * c, m, d)  # row = b*C + cap (model.py:805,822)
  sims = _SimsFn.apply(txt, vid_embds, text_weights.reshape(b * c, m), vid_weights.reshape(b, m))
  if c > 1:
    if merge == 'avg':
      sims = sims.view(b, c, b).mean(1)
    elif merge != 'indep':
      raise ValueError('unrecognised merge mode: {}'.format(merge))
  return sims.to(dev) if on_cpu else sims


def sharded_cross_view_inner_product(vid_embds, text_embds, vid_weights, text_weights, subspaces,
                                     merge_caption_similiarities='avg'):
  """model/model.py:789-837, same signature (dicts keyed by modality)."""
  vid = torch.stack([vid_embds[mod] for mod in subspaces], 1)
  b = vid.shape[0]
  txt = torch.stack([text_embds[mod].reshape(b, -1, vid.shape[-1]) for mod in subspaces], 1)  # (B,M,C,d)
  c = txt.shape[2]
  return cross_view_similarity(vid, txt, vid_weights.reshape(b, -1), text_weights.reshape(b, c, -1),
                               merge_caption_similiarities)


class _VideoPlan:
  """Per-(B,T) device buffers of the token pipeline."""
  KSPLIT = 768  # input channels per ReduceDim GEMM problem

  def __init__(self, net, bsz, t, device):
    mods, dims = net.modalities, net.expert_dims
    m = len(mods)
    self.batch, self.tokens, self.seq = bsz, t, 1 + m * (t + 1)
    self.rows = bsz * self.seq
    self.rows_alloc = ops.pad_rows(self.rows)
    self.generation = 0
    self.front_done = -1  # generation whose plan / cast launches rode along with the text heads (CENet._video_front)
    i32 = dict(device=device, dtype=torch.int32)
    self.counts = torch.zeros(bsz, **i32)
    self.cu = torch.zeros(bsz + 1, **i32)
    self.n_rows = torch.zeros(1, **i32)
    self.slot = torch.zeros(self.rows, **i32)
    self.row_index = torch.zeros(self.rows_alloc, **i32)
    self.type_ids = torch.zeros(self.rows_alloc, **i32)
    self.pos_ids = torch.zeros(self.rows_alloc, **i32)
    self.mask_bias = torch.zeros(self.rows_alloc, device=device, dtype=torch.float32)
    self.agg_row = torch.zeros(bsz * m, **i32)
    # source-row compaction (MmtVideoSrc): the ReduceDim projections only see live rows
    self.src_row = torch.zeros(self.rows_alloc, **i32)
    self.src_cnt = torch.zeros(m, **i32)
    self.xsrc = torch.zeros(m, bsz * t, **i32)
    self.src = MmtVideoSrc()
    self.src.src_row, self.src.src_cnt, self.src.xsrc = (x.data_ptr() for x in (self.src_row, self.src_cnt, self.xsrc))
    self.compact_rows = None
    self.features = None
    self.src_rows = bsz * (t + 1)
    self.src_rows_pad = _round_up(self.src_rows, 128)
    d = net.same_dim
    self.x, self.y, self.dy, self.y_part = {}, {}, {}, {}
    self.zero_bias = torch.zeros(d, device=device, dtype=torch.float32)
    chunks = {mod: min(3, -(-_round_up(dims[mod]['dim'], 128) // self.KSPLIT)) for mod in mods}
    if sum(chunks.values()) > 16:  # MMT_GEMM_GROUP_MAX problems per grouped launch
      chunks = {mod: 1 for mod in mods}
    self.xin = {}        # X_e actually read this step: self.x (filled by the cast kernel) or the RaggedFeatures' own
    self.precast = False
    for mod in mods:
      dpad = _round_up(dims[mod]['dim'], 128)
      self.x[mod] = torch.zeros(self.src_rows_pad, dpad, device=device, dtype=torch.bfloat16)
      self.y[mod] = torch.zeros(self.src_rows_pad, d, device=device, dtype=torch.float32)
      self.dy[mod] = torch.zeros(self.src_rows_pad, d, device=device, dtype=torch.bfloat16)
      self.y_part[mod] = [torch.zeros(self.src_rows_pad, d, device=device, dtype=torch.float32)
                          for _ in range(chunks[mod] - 1)]
    self.io = (MmtExpertIO * m)()
    self.inputs = None  # keeps the input tensors of the current batch alive


class CENet(nn.Module):
  """Whole cross-modal architecture (reference: model/model.py:45-661)."""

  def __init__(self, l2renorm, expert_dims, tokenizer, keep_missing_modalities, test_caption_mode,
               freeze_weights=False, mimic_ce_dims=False, concat_experts=False, concat_mix_experts=False,
               use_experts='origfeat', txt_inp=None, txt_agg=None, txt_pro=None, txt_wgh=None, vid_inp=None,
               vid_cont=None, vid_wgh=None, pos_enc=None, out_tok=None, use_mask='nomask', same_dim=512,
               vid_bert_params=None, txt_bert_params=None, agg_dims=None, normalize_experts=True,
               txt_bert=None, pack_tokens=True):
    super().__init__()
    self.expert_dims = expert_dims
    self.modalities = list(expert_dims.keys())
    self.test_caption_mode = test_caption_mode
    self.keep_missing_modalities = keep_missing_modalities
    self.l2renorm, self.same_dim = l2renorm, same_dim
    self.txt_inp, self.txt_agg, self.txt_pro, self.txt_wgh = txt_inp, txt_agg, txt_pro, txt_wgh
    self.vid_inp, self.vid_cont, self.vid_wgh = vid_inp, vid_cont, vid_wgh
    self.pos_enc, self.out_tok = pos_enc, out_tok
    self.vid_bert_params = vid_bert_params
    self.normalize_experts = normalize_experts
    self.pack_tokens = pack_tokens
    # the read-out only uses the AGG rows (model.py:583-587): let the last layer compute just those (exact)
    self.tail_rows_only = True
    self._vid_weights = {}
    # packed token rows of the NEXT minibatch as the loader counted them (None = unknown): lets the GEMM dispatcher price
    # a packed launch at its live size (include/mmt_hip.h: MmtBertBatch.live_rows_hint); `count_live_rows` computes it
    self.live_rows_hint = None
    self.text_live_rows_hint = None  # the same for the native text tower: real caption tokens (`count_live_tokens`)
    self.overlap_text_heads = False  # measured: no gain on MI355X (1.76 vs 1.74 ms/step), kept as an option
    self._side_streams = {}
    # the reference indexes nn.Embedding tables with these and raises IndexError when they do not fit; the kernels would
    # read out of bounds silently, so the check happens here
    type_vocab = (vid_bert_params or {}).get('type_vocab_size')
    if type_vocab is not None:
      bad = [m for m, e in expert_dims.items() if not 0 <= e['idx'] < type_vocab]
      if bad:
        raise IndexError('expert type ids of %r do not fit vid_bert_params.type_vocab_size = %d' % (bad, type_vocab))
    unsupported = []
    if vid_cont != 'bert': unsupported.append('vid_cont=%r' % vid_cont)
    if vid_inp != 'both': unsupported.append('vid_inp=%r' % vid_inp)
    if out_tok != 'mxp': unsupported.append('out_tok=%r' % out_tok)
    if pos_enc not in ('tint', 'none'): unsupported.append('pos_enc=%r' % pos_enc)
    if vid_wgh != 'none': unsupported.append('vid_wgh=%r' % vid_wgh)
    if not keep_missing_modalities: unsupported.append('keep_missing_modalities=False')
    if not normalize_experts: unsupported.append('normalize_experts=False')
    if txt_pro not in ('gbn', 'gem', 'lin'): unsupported.append('txt_pro=%r' % txt_pro)
    if txt_wgh not in ('emb', 'none'): unsupported.append('txt_wgh=%r' % txt_wgh)
    if unsupported:
      raise NotImplementedError('mmt_amd.CENet implements the published MMT configuration natively; '
                                'unsupported: ' + ', '.join(unsupported))
    if len(self.modalities) > 16:
      raise NotImplementedError('at most 16 experts')

    self.video_dim_reduce = nn.ModuleDict(
        {mod: ReduceDim(expert_dims[mod]['dim'], same_dim) for mod in self.modalities})
    self.vid_bert = BertModel(types.SimpleNamespace(**vid_bert_params))
    self.vid_bert.compute_pooler = False
    if self.vid_bert.config.hidden_size != same_dim:
      raise ValueError('vid_bert hidden_size must equal same_dim')

    # --- text tower (third party; model/model.py:136-190) ---
    if not (txt_agg or '').startswith('bert'):
      raise NotImplementedError('txt_agg=%r: only the BERT text tower of the published configs' % txt_agg)
    z = re.match(r'bert([a-z]{3})(\d*)(\D*)', txt_agg)
    assert z
    state, freeze_until = z.groups()[0], z.groups()[1]
    self.post_agg = z.groups()[2] if z.groups()[2] and z.groups()[2] != 'cls' else 'cls'
    if txt_bert_params is None:
      dout = vid_bert_params['hidden_dropout_prob']
      txt_bert_params = {'hidden_dropout_prob': dout, 'attention_probs_dropout_prob': dout}
    if txt_bert is None:
      # model/model.py:39,152-162: pretrained bert-base-cased (needs the HuggingFace cache or network), moved onto
      # the native engine; txt_bert='native' builds the same architecture with random weights (offline use: the
      # weights then come from a trained MMT checkpoint via load_state_dict)
      from transformers import BertModel as TxtBertModel
      from .text_bert import TextBertModel
      txt_bert = TextBertModel.from_hf(TxtBertModel.from_pretrained('bert-base-cased', **txt_bert_params))
    elif isinstance(txt_bert, str):
      if txt_bert != 'native':
        raise ValueError("txt_bert: a module, None (pretrained bert-base-cased) or 'native' (random init)")
      from .text_bert import TextBertModel, bert_base_cased_config
      txt_bert = TextBertModel(bert_base_cased_config(**txt_bert_params))
    self.txt_bert = txt_bert
    self._native_text_tower = hasattr(txt_bert, 'flat_named_params')
    if self._native_text_tower:
      txt_bert.cls_only = self.post_agg == 'cls'  # model/model.py:378-379 reads last_layer[:, 0] only
    if state == 'frz':
      for name, param in self.txt_bert.named_parameters():
        parts = name.split('.')
        if parts[0] != 'encoder':
          continue
        if freeze_until:
          if len(parts) > 2 and parts[2].isdigit() and int(parts[2]) < int(freeze_until):
            param.requires_grad = False
        else:
          param.requires_grad = False
    if txt_inp == 'bertfrz':
      for param in self.txt_bert.embeddings.parameters():
        param.requires_grad = False
    text_dim = self.txt_bert.config.hidden_size

    self.text_GU = nn.ModuleDict()
    for mod in self.modalities:
      if txt_pro == 'gbn':
        self.text_GU[mod] = GatedEmbeddingUnit(text_dim, same_dim, use_bn=True, normalize=normalize_experts)
      elif txt_pro == 'gem':
        self.text_GU[mod] = GatedEmbeddingUnit(text_dim, same_dim, use_bn=False, normalize=normalize_experts)
      else:
        self.text_GU[mod] = ReduceDim(text_dim, same_dim)
    if txt_wgh == 'emb':
      self.moe_fc_txt = nn.ModuleDict({mod: nn.Linear(text_dim, 1) for mod in self.modalities})
      self.moe_txt_dropout = nn.Dropout(txt_bert_params['hidden_dropout_prob'])

    # --- flat storage of everything the engine touches ---
    # Layout order = reverse of the order in which the backward finishes the gradients:
    #   [expert projections | embeddings | layer 0 | ... | layer L-1 | text heads]
    # so the spans that a data-parallel step reduces early (text heads + top layer) and last (layer 0 + embeddings +
    # projections) are each ONE contiguous run (grad_regions()).
    named = []
    for mod in self.modalities:
      named += [('video_dim_reduce.%s.fc.weight' % mod, self.video_dim_reduce[mod].fc.weight),
                ('video_dim_reduce.%s.fc.bias' % mod, self.video_dim_reduce[mod].fc.bias)]
    named += self.vid_bert.engine_named_params('vid_bert.')
    self._native_text_heads = txt_pro in ('gbn', 'gem')
    if self._native_text_heads:
      named += [('text_GU.%s.fc.weight' % mod, self.text_GU[mod].fc.weight) for mod in self.modalities]  # contiguous
      for mod in self.modalities:
        gu = self.text_GU[mod]
        named += [('text_GU.%s.fc.bias' % mod, gu.fc.bias), ('text_GU.%s.cg.fc.weight' % mod, gu.cg.fc.weight),
                  ('text_GU.%s.cg.fc.bias' % mod, gu.cg.fc.bias),
                  ('text_GU.%s.cg.batch_norm.weight' % mod, gu.cg.batch_norm.weight),
                  ('text_GU.%s.cg.batch_norm.bias' % mod, gu.cg.batch_norm.bias)]
      if txt_wgh == 'emb':
        for mod in self.modalities:
          named += [('moe_fc_txt.%s.weight' % mod, self.moe_fc_txt[mod].weight),
                    ('moe_fc_txt.%s.bias' % mod, self.moe_fc_txt[mod].bias)]
    self._flat = FlatParams(named)
    self._th_generation = 0
    self._th_ws = {}
    self._nbt = None
    self.vid_bert.register_shadows(self._flat)
    for mod in self.modalities:
      dim = expert_dims[mod]['dim']
      self._flat.add_shadow(('reduce', mod), [self.video_dim_reduce[mod].fc.weight], same_dim, dim,
                            k_pad=_round_up(dim, 128))
    self.vid_bert.attach_flat(self._flat)
    self._plans = {}
    self._stages = None

  def __str__(self):
    n = sum(p.numel() for p in self.parameters() if p.requires_grad)
    return super().__str__() + '\nTrainable parameters: {}'.format(n)

  def grad_regions(self, split_bottom=False):
    """Contiguous (offset, count) spans of the flat gradient buffer in the order the backward finishes them:
    [('top', text heads + last layer), ('layer<L-2>', ...), ..., ('layer1', ...),
     ('bottom', expert projections + embeddings + layer 0)]  (one layer: 'top' = text heads only).
    split_bottom: 'bottom' as ('layer0', layer 0) + ('bottom', expert projections + embeddings) -- layer 0's weight
    gradients are final before the embedding / token stage runs, so a data-parallel step can already reduce them."""
    f, vb = self._flat, self.vid_bert
    n_layers = vb.config.num_hidden_layers
    named = vb.engine_named_params()
    per_layer = [[p for n, p in named if n.startswith('encoder.layer.%d.' % l)] for l in range(n_layers)]
    emb = [p for n, p in named if n.startswith('embeddings.')]
    top = list(self._text_head_params()) if self._native_text_heads else []
    if n_layers >= 2:
      top = per_layer[n_layers - 1] + top
    out = [('top', f.span(top))] if top else []
    for l in range(n_layers - 2, 0, -1):
      out.append(('layer%d' % l, f.span(per_layer[l])))
    if split_bottom and n_layers >= 2:
      out.append(('layer0', f.span(per_layer[0])))
      out.append(('bottom', f.span(self._reduce_params() + emb)))
    else:
      out.append(('bottom', f.span(self._reduce_params() + emb + per_layer[0])))
    return out

  @staticmethod
  def count_live_rows(features_ind):
    """Packed token rows of a minibatch from its HOST-side indicator arrays {expert: (B, T) 0/1} (what the reference's
    collate hands over, base/base_dataset.py:779-806): one CLS row per sample + per expert one AGG row and its valid
    feature rows.  The loader calls this before the upload and assigns the result to `live_rows_hint`."""
    first = next(iter(features_ind.values()))
    total = int(first.shape[0])
    for ind in features_ind.values():
      total += int(ind.shape[0]) + int((torch.as_tensor(ind) != 0).sum())
    return total

  @staticmethod
  def count_live_tokens(token_ids):
    """Real caption tokens of a minibatch from its HOST-side `token_ids` (B, C, W, 2) = [id, valid] (model/model.py:353-357
    reads the same two planes): what the loader assigns to `text_live_rows_hint`."""
    return int((torch.as_tensor(token_ids)[..., 1] != 0).sum())

  def engine_params(self):
    """Parameters living in the flat buffer (video side), in layout order."""
    return self._flat.params

  def flats(self):
    """Every flat parameter buffer of the model: the video side (+ text heads) and, when it runs on the native
    engine, the text tower's own (its 108 M parameters are an order of magnitude more than the rest)."""
    out = [self._flat]
    if self._native_text_tower:
      out.append(self.txt_bert.build_flat())
    return out

  # ---- video side --------------------------------------------------------------------------------
  def _reduce_params(self):
    out = []
    for mod in self.modalities:
      out += [self.video_dim_reduce[mod].fc.weight, self.video_dim_reduce[mod].fc.bias]
    return out

  def _prepare(self, device):
    if self._flat.ensure(device):
      self.vid_bert._structs = {}
      self._th_ws = {}
    self._flat.pack()  # no-op unless a weight changed (version counters / FlatAdam's dirty flag)
    if torch.is_grad_enabled():
      self._flat.select_grad_buffer()
    self.vid_bert._ensure_ready(device)

  def _video_tokens_forward(self, plan):
    L, m, d = _lib.lib(), len(self.modalities), self.same_dim
    io, stream = plan.io, ops._stream()
    max_pos = self.vid_bert_params['max_position_embeddings'] - 1
    # a training forward draws a fresh dropout seed: the plan kernel bumps the encoder's seed word itself (and the
    # encoder is told not to), one launch less per step
    vb = self.vid_bert
    bump = vb._seed_dev if (plan.bump_seed and vb._seed_dev is not None) else None
    vb._seed_bumped = bump is not None
    if plan.front_done != plan.generation:  # (else: plan and cast rode along with the text heads' launches, _video_front)
      check(L.mmt_video_plan(io, m, plan.batch, plan.tokens, int(self.pack_tokens), max_pos, ops._p(plan.counts),
                             ops._p(plan.cu), ops._p(plan.n_rows), ops._p(plan.slot), ops._p(plan.row_index),
                             ops._p(plan.type_ids), ops._p(plan.pos_ids), ops._p(plan.mask_bias), ops._p(plan.agg_row),
                             ops._p(bump), ctypes.byref(plan.src), stream), 'mmt_video_plan')
      if not plan.precast:
        check(L.mmt_video_cast(io, m, plan.batch, plan.tokens, ctypes.byref(plan.src), stream), 'mmt_video_cast')
    # wide experts are cut along K into chunks of <= KSPLIT columns (own tiles, partial products summed by the scatter
    # kernel): the launch lasts as long as its longest K loop, and rgb / scene have 2048 / 2208 input channels
    items, index = [], []
    for i, mod in enumerate(self.modalities):
      x, w = plan.xin[mod], self._flat.shadow(('reduce', mod))[0]
      outs = [plan.y[mod]] + plan.y_part[mod]
      step = _round_up(-(-x.shape[1] // len(outs)), 64)  # even chunks, whole K steps of the GEMM
      for c, out in enumerate(outs):
        k0 = c * step
        k1 = x.shape[1] if c == len(outs) - 1 else k0 + step
        items.append((x[:, k0:k1], w[:, k0:k1], out, self.video_dim_reduce[mod].fc.bias if c == 0 else plan.zero_bias))
        index.append(i)
    ops.gemm_nt_grouped(items, m=plan.src_rows, n_rows_dev=plan.src_cnt, n_rows_index=index)
    feats = torch.empty(plan.rows_alloc, d, device=plan.slot.device, dtype=torch.float32)
    check(L.mmt_video_scatter(io, m, plan.batch, plan.tokens, d, ops._p(plan.n_rows), ops._p(plan.row_index),
                              ctypes.byref(plan.src), ops._p(feats), stream), 'mmt_video_scatter')
    return feats

  def _video_front(self, plan):
    """The plan (+ cast) launches of this forward as a descriptor the text heads' first two launches carry as extra blocks
    (MmtVideoFront, texthead2.hip): two dependent launches less per step."""
    vb = self.vid_bert
    bump = vb._seed_dev if (plan.bump_seed and vb._seed_dev is not None) else None
    f = MmtVideoFront()
    f.experts = ctypes.addressof(plan.io)
    f.M, f.B, f.T, f.pack = len(self.modalities), plan.batch, plan.tokens, int(self.pack_tokens)
    f.max_pos, f.do_cast = self.vid_bert_params['max_position_embeddings'] - 1, int(not plan.precast)
    for name, t in (('counts', plan.counts), ('cu_seqlens', plan.cu), ('n_rows_dev', plan.n_rows), ('slot', plan.slot),
                    ('row_index', plan.row_index), ('type_ids', plan.type_ids), ('pos_ids', plan.pos_ids),
                    ('mask_bias', plan.mask_bias), ('agg_row', plan.agg_row)):
      setattr(f, name, t.data_ptr())
    f.seed_bump = bump.data_ptr() if bump is not None else None
    f.src = ctypes.addressof(plan.src)
    return f

  def _video_tokens_backward(self, plan, dfeat, side_stream=None):
    """side_stream: the ReduceDim weight gradients (the last kernel of the backward; only the optimizer reads them) go
    there, ordered after the scatter; the caller joins."""
    L, m, d = _lib.lib(), len(self.modalities), self.same_dim
    stream = ops._stream()
    check(L.mmt_video_scatter_bwd(plan.io, m, plan.batch, plan.tokens, d, ops._p(plan.n_rows), ops._p(plan.row_index),
                                  ctypes.byref(plan.src), ops._p(dfeat), stream), 'mmt_video_scatter_bwd')
    grad_buf = self._flat.current_grad()
    grads, items = [], []
    for mod in self.modalities:  # every ReduceDim weight + bias gradient in ONE grouped launch
      fc = self.video_dim_reduce[mod].fc
      gw, gb = self._flat.view(fc.weight, grad_buf), self._flat.view(fc.bias, grad_buf)
      items.append((plan.dy[mod], plan.xin[mod], gw, gb))
      grads += [gw if fc.weight.requires_grad else None, gb if fc.bias.requires_grad else None]
    if side_stream is not None:
      side_stream.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side_stream):
        ops.wgrad_grouped(items, plan.src_rows, item_rows_dev=plan.src_cnt)
    else:
      ops.wgrad_grouped(items, plan.src_rows, item_rows_dev=plan.src_cnt)
    return grads

  def video_embeddings(self, features, features_t, features_ind, features_maxpool, plan=None):
    """(B, M, d) L2-normalised expert embeddings of the video side (model.py:426-437, 485-587, 621-625).
    plan: what `_video_prepare` returned for these inputs (forward() prepares it early, so that the text heads' launches
    can carry the plan and the cast)."""
    if plan is None:
      plan = self._video_prepare(features, features_t, features_ind, features_maxpool)
    mods, bsz, dev = self.modalities, plan.batch, plan.slot.device
    feats = _VideoTokensFn.apply(self, plan, *self._reduce_params())
    batch = EngineBatch(None, plan.type_ids, plan.pos_ids if self.pos_enc != 'none' else None, plan.mask_bias,
                        plan.rows, bsz, plan.seq, cu_seqlens=plan.cu,
                        row_index=plan.row_index if self.pack_tokens else None,  # dense rows ARE the original coordinates
                        n_rows_dev=plan.n_rows if self.pack_tokens else None,
                        out_rows=plan.agg_row if self.tail_rows_only else None, n_out_per_sample=len(mods))
    # the live token rows as the HOST knows them (tile choice only; the kernels read the device count): set by the loader
    # (`live_rows_hint`, e.g. RaggedCollator's count), or read off the wire format, which carries them
    hint = self.live_rows_hint
    if hint is None and isinstance(features, RaggedFeatures):
      hint = bsz + sum(int(v) for v in features.live.values())  # CLS + per expert (AGG + valid feature rows)
    batch.live_rows_hint = int(hint or 0) if self.pack_tokens else 0
    last = self.vid_bert.run_engine(batch, feats)
    # handles for callers that drive the backward of this forward stage by stage (train_step.GraphedTrainStep)
    self._stages = dict(plan=plan, feats=feats, batch=batch, last=last) if last.requires_grad else None
    if self.vid_bert.compact_output(batch, plan.rows_alloc):  # the engine returned just the AGG rows, in agg_row order
      if plan.compact_rows is None:
        plan.compact_rows = torch.arange(bsz * len(mods), device=dev, dtype=torch.int32)
      vid = _ReadoutFn.apply(last, plan.compact_rows, bsz * len(mods), True, self._stages)
    else:
      vid = _ReadoutFn.apply(last, plan.agg_row, bsz * len(mods), False, self._stages)
    vid = vid.view(bsz, len(mods), self.same_dim)
    if self._stages is not None:
      self._stages['vid_embds'] = vid
    return vid

  def _video_prepare(self, features, features_t, features_ind, features_maxpool):
    """Host side of the video forward: checks the inputs, binds them to the (B, T) plan's expert table.  No launches."""
    mods = self.modalities
    ragged = features if isinstance(features, RaggedFeatures) else None
    if ragged is not None:
      # wire format of feature_store.py: X_e arrives compact and in bf16, the cast kernel has nothing to do
      if not self.pack_tokens:
        raise NotImplementedError('RaggedFeatures carry no padded rows: they need pack_tokens=True')
      if [n for n, _ in ragged.layout.experts] != mods or any(
          d != self.expert_dims[n]['dim'] for n, d in ragged.layout.experts):
        raise ValueError('RaggedFeatures experts %r do not match the model\'s %r' % (ragged.layout.experts, mods))
      dev, bsz, t = ragged.device, ragged.batch, ragged.tokens
      if dev.type != 'cuda':
        raise RuntimeError('mmt_amd.CENet runs on the GPU only (no CPU fallback)')
      features_ind, features_t = ragged.ind, ragged.t
    else:
      f0 = features[mods[0]]
      if not f0.is_cuda:
        raise RuntimeError('mmt_amd.CENet runs on the GPU only (no CPU fallback)')
      dev = f0.device
      bsz, t = f0.shape[0], f0.shape[1]
      for mod in mods:
        if features[mod].shape[1] != t:
          raise NotImplementedError('all experts must share max_expert_tokens (as in every published config)')
    self._prepare(dev)
    key = (bsz, t, dev)
    plan = self._plans.get(key)
    if plan is None:
      plan = self._plans[key] = _VideoPlan(self, bsz, t, dev)
    plan.generation += 1
    keep = []
    for i, mod in enumerate(mods):
      io = plan.io[i]
      if ragged is not None:
        tensors = [x.detach().to(device=dev, dtype=torch.float32).contiguous() for x in (features_ind[mod], features_t[mod])]
        x = ragged.x[mod]
        if x.shape != plan.x[mod].shape:
          raise ValueError('RaggedFeatures layout does not match the model (rows/K padding)')
        keep.append(tensors + [x])
        io.feat = io.maxpool = None
        io.ind, io.t = (v.data_ptr() for v in tensors)
      else:
        tensors = [features[mod], features_maxpool[mod], features_ind[mod], features_t[mod]]
        tensors = [x.detach().to(device=dev, dtype=torch.float32).contiguous() for x in tensors]
        keep.append(tensors)
        x = plan.x[mod]
        io.feat, io.maxpool, io.ind, io.t = (v.data_ptr() for v in tensors)
      plan.xin[mod] = x
      io.x, io.y, io.dy = x.data_ptr(), plan.y[mod].data_ptr(), plan.dy[mod].data_ptr()
      io.D, io.Dpad = self.expert_dims[mod]['dim'], x.shape[1]
      io.n_part = len(plan.y_part[mod])
      for c, part in enumerate(plan.y_part[mod]):
        io.y_part[c] = part.data_ptr()
      io.type_idx, io.rows_pad = self.expert_dims[mod]['idx'], plan.src_rows_pad
    plan.inputs = keep
    plan.precast = ragged is not None
    # (grad mode is off inside autograd.Function.forward: decide here whether this forward draws a fresh dropout seed)
    plan.bump_seed = self.vid_bert.training and torch.is_grad_enabled()
    return plan

  # ---- text heads (native) -------------------------------------------------------------------------
  def _text_head_params(self):
    out = [self.text_GU[mod].fc.weight for mod in self.modalities]
    for mod in self.modalities:
      gu = self.text_GU[mod]
      out += [gu.fc.bias, gu.cg.fc.weight, gu.cg.fc.bias, gu.cg.batch_norm.weight, gu.cg.batch_norm.bias]
    if self.txt_wgh == 'emb':
      for mod in self.modalities:
        out += [self.moe_fc_txt[mod].weight, self.moe_fc_txt[mod].bias]
    return out

  def _text_heads_struct(self, grad_buf):
    f, h = self._flat, MmtTextHeads()
    for i, mod in enumerate(self.modalities):
      gu, bn = self.text_GU[mod], self.text_GU[mod].cg.batch_norm
      pairs = dict(w1=gu.fc.weight, b1=gu.fc.bias, w2=gu.cg.fc.weight, b2=gu.cg.fc.bias, bn_gamma=bn.weight,
                   bn_beta=bn.bias)
      if self.txt_wgh == 'emb':
        pairs.update(moe_w=self.moe_fc_txt[mod].weight, moe_b=self.moe_fc_txt[mod].bias)
      for name, p in pairs.items():
        getattr(h, name)[i] = f.ptr(p)
        if grad_buf is not None and p.requires_grad:
          getattr(h, 'g_' + name)[i] = f.ptr(p, grad_buf)
      h.running_mean[i] = bn.running_mean.data_ptr()
      h.running_var[i] = bn.running_var.data_ptr()
    return h

  def _text_heads_fast(self, n, k):
    """True iff the small-batch kernels (3 + 3 launches, texthead2.hip) serve n caption rows of width k."""
    return bool(_lib.lib().mmt_text_heads_fast(n, len(self.modalities), self.same_dim, k))

  def _text_heads_opts(self, moe_drop_p, nbt):
    o = MmtTextHeadsOpts()
    if moe_drop_p > 0.0:
      if self._th_key is None or self._th_key.device != self.vid_bert._seed_dev.device:
        self._th_key = torch.zeros(1, dtype=torch.int32, device=self.vid_bert._seed_dev.device)
      o.moe_drop_key = _MoeDropoutFn.SITE_KEY
      o.moe_drop_thr16, o.moe_drop_scale = ops.dropout_params(moe_drop_p)
      o.seed_dev, o.key_dev = self.vid_bert._seed_dev.data_ptr(), self._th_key.data_ptr()
    if nbt is not None:
      o.num_batches_tracked = nbt.data_ptr()
    return o

  _th_key = None
  _pending_front = None  # video plan whose plan / cast launches the next text-heads forward carries along
  _th_front = None
  # lab switch (same-box A/B): MMT_FRONT_FUSE=0 keeps the plan and the cast as launches of their own
  front_fuse = os.environ.get('MMT_FRONT_FUSE', '1') != '0'

  def _text_heads_forward(self, text, text_moe, caps, moe_drop_p=0.0):
    n, k = text.shape
    m, d = len(self.modalities), self.same_dim
    L = _lib.lib()
    key = (n, text.device)
    ws = self._th_ws.get(key)
    if ws is None:
      ws = self._th_ws[key] = torch.zeros(L.mmt_text_heads_workspace_floats(n, m, d), device=text.device)
    self._th_generation += 1
    self._th_text, self._th_text_moe = text, text_moe
    use_bn = int(self.txt_pro == 'gbn')
    fast = self._text_heads_fast(n, k)
    if moe_drop_p > 0.0 and not fast:
      raise RuntimeError('on-the-fly MoE dropout needs the small-batch text-head kernels')
    nbt = None
    if self.training and use_bn:
      if n <= 1:
        raise ValueError('Expected more than 1 value per channel when training')  # as nn.BatchNorm1d does
      if self._nbt is None or self._nbt.device != text.device or any(
          self.text_GU[mod].cg.batch_norm.num_batches_tracked.data_ptr() != self._nbt[i].data_ptr()
          for i, mod in enumerate(self.modalities)):
        vals = [int(self.text_GU[mod].cg.batch_norm.num_batches_tracked) for mod in self.modalities]
        self._nbt = torch.tensor(vals, dtype=torch.long, device=text.device)
        for i, mod in enumerate(self.modalities):
          self.text_GU[mod].cg.batch_norm.num_batches_tracked.data = self._nbt[i]
      if fast:
        nbt = self._nbt  # incremented by the BatchNorm kernel itself (one launch less per step)
      else:
        self._nbt.add_(1)
    embds = torch.empty(n // caps, m, caps, d, device=text.device, dtype=torch.float32)
    tw = torch.empty(n // caps, caps, m, device=text.device, dtype=torch.float32) if self.txt_wgh == 'emb' else None
    h = self._text_heads_struct(None)
    self._th_opts = self._text_heads_opts(moe_drop_p, nbt)
    front_plan, self._pending_front = self._pending_front, None
    if front_plan is not None and fast:
      self._th_front = self._video_front(front_plan)  # (kept alive: the opts hold its address)
      self._th_opts.video_front = ctypes.addressof(self._th_front)
      front_plan.front_done = front_plan.generation
    check(L.mmt_text_heads_fwd(ctypes.byref(h), ops._p(text), ops._p(text_moe), n, caps, m, d, k, use_bn, int(self.training), ops._p(ws),
                               ops._p(embds), ops._p(tw), ctypes.byref(self._th_opts), ops._stream()), 'mmt_text_heads_fwd')
    if tw is None:
      tw = torch.full((n // caps, caps, m), 1.0 / m, device=text.device)  # ones, L1-normalised (model.py:612,618)
    self._th_tw = tw
    return embds, tw

  def _text_heads_backward(self, caps, de, dtw, need_dtext, need_dmoe, training):
    text, text_moe = self._th_text, self._th_text_moe
    n, k = text.shape
    m, d = len(self.modalities), self.same_dim
    L = _lib.lib()
    grad_buf = self._flat.current_grad()
    h = self._text_heads_struct(grad_buf)
    dtext = torch.empty_like(text) if need_dtext else None
    dmoe = torch.empty_like(text) if (need_dmoe and text_moe is not None) else None
    w1_all = self._flat.ptr(self.text_GU[self.modalities[0]].fc.weight)
    has_moe = self.txt_wgh == 'emb'
    opts = self._th_opts
    fused_drop = opts.moe_drop_thr16 != 0 and text_moe is None
    # on-the-fly MoE dropout: the (masked) gradient through the MoE branch arrives in its own buffer and joins dtext
    dmoe_fused = torch.empty_like(text) if (fused_drop and need_dtext and has_moe) else None
    bwd_opts = MmtTextHeadsOpts.from_buffer_copy(opts)
    bwd_opts.num_batches_tracked = None
    bwd_opts.video_front = None
    check(L.mmt_text_heads_bwd(ctypes.byref(h), ops._p(text), ops._p(text_moe), ctypes.c_void_p(w1_all), n, caps, m, d, k,
                               int(self.txt_pro == 'gbn'), int(training), ops._p(self._th_ws[(n, text.device)]),
                               ops._p(de.contiguous()), ops._p(self._th_tw) if has_moe else None,
                               ops._p(dtw.contiguous()) if has_moe and dtw is not None else None, ops._p(dtext),
                               ops._p(dmoe_fused if dmoe_fused is not None else dmoe), ctypes.byref(bwd_opts), ops._stream()),
          'mmt_text_heads_bwd')
    if dmoe_fused is not None and dtw is not None:
      dtext.add_(dmoe_fused)
    grads = [self._flat.view(p, grad_buf) if p.requires_grad else None for p in self._text_head_params()]
    return dtext, dmoe, grads

  # ---- text side: token ids -> text tower (native engine, or a foreign module) -> [CLS] / pooled features -------------
  def text_features(self, token_ids, device):
    """model/model.py:349-379: (B, C, W, 2) -> (B*C, text_dim) via the text tower's [CLS]."""
    b, c, w, f = token_ids.size()
    if getattr(self.txt_bert, 'ignores_token_inputs', False):  # a precomputed-feature stand-in: skip the id/mask prep
      out = self.txt_bert(None)
      return out[0][:, 0] if self.post_agg == 'cls' else (torch.max(out[0][:, 1:], 1)[0] if self.post_agg == 'mxp'
                                                          else torch.mean(out[0][:, 1:], 1))
    if self._native_text_tower:  # (the loader's token count, or -- token_ids still on the host -- counted here)
      hint = self.text_live_rows_hint
      if hint is None and not token_ids.is_cuda:
        hint = self.count_live_tokens(token_ids)
      self.txt_bert.live_rows_hint = hint
    tok = token_ids.view(b * c, w, f).to(device)
    input_ids = tok[:, :, 0].long()
    attention_mask = tok[:, :, 1].long()
    position_ids = torch.arange(w, device=device).unsqueeze(0).expand(b * c, w)
    token_type_ids = torch.zeros_like(input_ids)
    out = self.txt_bert(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids,
                        position_ids=position_ids, head_mask=None)
    last_layer = out[0]
    if self.post_agg == 'cls':
      return last_layer[:, 0]
    if self.post_agg == 'mxp':
      return torch.max(last_layer[:, 1:], 1)[0]
    return torch.mean(last_layer[:, 1:], 1)

  def compute_weights_from_emb(self, embd):
    """model/model.py:262-283 (text branch)."""
    embd = self.moe_txt_dropout(embd)
    b, k, d = embd.size()
    flat = embd.view(b * k, d)
    w = torch.cat([self.moe_fc_txt[mod](flat) for mod in self.modalities], dim=-1)
    return F.softmax(w, dim=1).view(b, k, len(self.modalities))

  def forward(self, token_ids, features, features_t, features_ind, features_avgpool, features_maxpool,
              query_masks, out='conf', device=None, debug=None):
    dev = (features.device if isinstance(features, RaggedFeatures) else features[self.modalities[0]].device) \
        if device is None else torch.device(device)
    if dev.type != 'cuda':
      raise RuntimeError('mmt_amd.CENet runs on the GPU only (no CPU fallback)')
    b, c = token_ids.size(0), token_ids.size(1)
    m = len(self.modalities)
    text = self.text_features(token_ids, dev)                                   # (B*C, text_dim)
    self._prepare(dev)
    plan = self._video_prepare(features, features_t, features_ind, features_maxpool)
    side = None
    self._pending_front = None
    if self._native_text_heads:
      # The text heads are ~15 tiny latency-bound launches that are independent of the video encoder until the
      # similarity: they run on a side stream (fork/join, also under graph capture) and hide under the encoder's GEMMs;
      # autograd runs their backward on the same side stream.
      cur = torch.cuda.current_stream(dev)
      if self.overlap_text_heads:
        side = self._side_streams.get(dev)
        if side is None:
          side = self._side_streams[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
      with torch.cuda.stream(side if side is not None else cur):
        text_moe, moe_drop_p = None, 0.0
        if self.txt_wgh == 'emb' and self.training and self.moe_txt_dropout.training and self.moe_txt_dropout.p > 0:
          # model.py:274: dropout only in front of the MoE logits
          if self.vid_bert._seed_dev is not None and self._text_heads_fast(text.shape[0], text.shape[1]):
            moe_drop_p = float(self.moe_txt_dropout.p)  # applied on the fly by the text-head kernels
          elif text.numel() % 4 == 0 and self.vid_bert._seed_dev is not None:
            text_moe = _MoeDropoutFn.apply(text, self.moe_txt_dropout.p, self.vid_bert._seed_dev)
          else:
            text_moe = self.moe_txt_dropout(text)
        if self.front_fuse and side is None:
          self._pending_front = plan  # (taken up only by the small-batch text-head kernels)
        text_embds, text_weights = _TextHeadsFn.apply(self, text, text_moe, c, moe_drop_p, *self._text_head_params())
        self._pending_front = None
    else:
      text_embd = [self.text_GU[mod](text).view(b, c, -1) for mod in self.modalities]  # model.py:413-417
      tv = text.view(b, c, -1)
      text_weights = self.compute_weights_from_emb(tv) if self.txt_wgh == 'emb' else torch.ones(b, c, m, device=dev)
      text_weights = F.normalize(text_weights, p=1, dim=-1)                       # model.py:618
      text_embds = torch.stack([F.normalize(t, dim=-1) for t in text_embd], 1)    # (B,M,C,d) model.py:623
    vid_embds = self.video_embeddings(features, features_t, features_ind, features_maxpool, plan=plan)
    if side is not None:
      cur.wait_stream(side)
      for t in (text_embds, text_weights):
        t.record_stream(cur)
    vw_key = (b, m, dev)
    vid_weights = self._vid_weights.get(vw_key)                                   # ones, L1-normalised model.py:594,607
    if vid_weights is None:
      vid_weights = self._vid_weights[vw_key] = torch.full((b, m), 1.0 / m, device=dev)
    merge = 'avg' if self.training else self.test_caption_mode                  # model.py:627-631
    self.merge_caption_similarities = merge
    if out == 'conf':
      return {'modalities': self.modalities,
              'cross_view_conf_matrix': cross_view_similarity(vid_embds, text_embds, vid_weights, text_weights, merge)}
    return {'vid_embds': vid_embds, 'text_embds': text_embds, 'vid_weights': vid_weights,
            'text_weights': text_weights}
