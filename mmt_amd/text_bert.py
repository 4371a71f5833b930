"""The text tower on the MI355X engine: a drop-in for the HuggingFace `BertModel` the reference fine-tunes
(`TxtBertModel.from_pretrained('bert-base-cased', ...)`, model/model.py:152-162; called at model/model.py:371-376 with
input_ids / attention_mask / token_type_ids / position_ids, and only `output[0]` is read).

It is the video-BERT engine with three differences: a word-embedding lookup in front (HF BertEmbeddings: word + position
+ token-type -> LayerNorm -> dropout), head dimension 64 (12 heads x 64 for BERT-base), and HuggingFace parameter names
(`LayerNorm` instead of the reference video-BERT's `layer_norm`) so that `bert-base-cased` checkpoints load.
Same numerics as the video side (bf16 MFMA operands, fp32 accumulation / residual stream / LayerNorm / softmax); same
C-ABI (mmt_bert_forward / mmt_bert_backward) plus mmt_rows_gather / mmt_embedding_grad for the lookup.  GPU only.

The arithmetic lives in a third-party dependency of the reference (transformers==3.1.0, requirements.txt:42) that is
not vendored; oracle/mmt_oracle.py:text_bert_model restates it and is pinned against the installed transformers'
BertModel by oracle/gen_golden.py (tests/golden/text_bert.npz).
"""
import types

import torch
from torch import nn

from . import _lib, ops
from ._lib import check
from .bert import BertModel, EngineBatch

_CFG_KEYS = ('vocab_size', 'hidden_size', 'num_hidden_layers', 'num_attention_heads', 'intermediate_size', 'hidden_act',
             'hidden_dropout_prob', 'attention_probs_dropout_prob', 'max_position_embeddings', 'type_vocab_size',
             'initializer_range', 'layer_norm_eps', 'pad_token_id')
_BERT_BASE_CASED = dict(vocab_size=28996, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                        intermediate_size=3072, hidden_act='gelu', hidden_dropout_prob=0.1,
                        attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                        initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0)


def bert_base_cased_config(**overrides):
  """The published bert-base-cased architecture (what model/model.py:152 asks from_pretrained for)."""
  cfg = dict(_BERT_BASE_CASED)
  cfg.update(overrides)
  return types.SimpleNamespace(**cfg)


class _WordEmbeddingFn(torch.autograd.Function):
  """features[r] = table[ids[r]] on the padded row grid; backward = deterministic scatter-add into the flat gradient."""

  @staticmethod
  def forward(ctx, model, ids_i32, rows, table, n_rows_dev=None):
    R, d = ids_i32.shape[0], table.shape[1]
    out = torch.zeros(R, d, device=table.device, dtype=torch.float32)
    check(_lib.lib().mmt_rows_gather(ops._p(table), ops._p(ids_i32), rows, d, ops._p(out), None, None, ops._stream()),
          'mmt_rows_gather')
    ctx.model, ctx.rows, ctx.n_rows_dev = model, rows, n_rows_dev
    ctx.save_for_backward(ids_i32)
    return out

  @staticmethod
  def backward(ctx, g):
    (ids_i32,) = ctx.saved_tensors
    model = ctx.model
    table = model.embeddings.word_embeddings.weight
    if not table.requires_grad:
      return None, None, None, None, None
    flat = model._flat
    gview = flat.view(table, flat.current_grad())
    gview.zero_()
    pad = model.embeddings.word_embeddings.padding_idx
    check(_lib.lib().mmt_embedding_grad(ops._p(g.contiguous()), ops._p(ids_i32), ctx.rows, table.shape[1], table.shape[0],
                                        -1 if pad is None else int(pad),
                                        ops._p(ctx.n_rows_dev) if ctx.n_rows_dev is not None else None, ops._p(gview),
                                        ops._stream()), 'mmt_embedding_grad')
    return None, None, None, gview, None


class TextBertModel(BertModel):
  r"""`TextBertModel(config)`: `config` is a HuggingFace `BertConfig` or any namespace with its fields
  (`bert_base_cased_config()`).  `forward(input_ids, attention_mask, token_type_ids, position_ids, head_mask=None)`
  -> `(sequence_output, pooled_output)` like transformers' BertModel.  With `cls_only = True` (CENet sets it for
  `post_agg == 'cls'`, model/model.py:378-379) only the [CLS] rows of the last layer are computed and
  `sequence_output` is (B, 1, hidden)."""

  def __init__(self, config):
    cfg = types.SimpleNamespace(**{k: getattr(config, k, _BERT_BASE_CASED[k]) for k in _CFG_KEYS})
    if cfg.hidden_size != cfg.num_attention_heads * 64 and cfg.hidden_size != cfg.num_attention_heads * 128:
      raise NotImplementedError('native text tower: head dimension 64 or 128')
    super().__init__(cfg)
    emb = nn.Embedding(cfg.vocab_size, cfg.hidden_size, padding_idx=cfg.pad_token_id)
    emb.weight.data.normal_(mean=0.0, std=cfg.initializer_range)
    if cfg.pad_token_id is not None:
      emb.weight.data[cfg.pad_token_id].zero_()
    self.embeddings.word_embeddings = emb
    self.cls_only = False
    # real (un-padded) caption tokens of the NEXT minibatch as the loader counted them (None = unknown): the GEMM dispatcher
    # prices the packed launches at their live size (MmtBertBatch.live_rows_hint; CENet.text_live_rows_hint forwards it)
    self.live_rows_hint = None
    self.pack_tokens = True      # with cls_only: drop the padded tokens (exact, see mmt_text_plan)
    self.compute_pooler = False  # model/model.py:376 reads output[0] only
    self._plans = {}
    self._register_load_state_dict_pre_hook(self._hf_names_in)
    self._register_state_dict_hook(self._hf_names_out)

  # ---- HuggingFace parameter names ------------------------------------------------------------------
  @staticmethod
  def _hf_names_in(state_dict, prefix, *args):
    for k in [k for k in state_dict if k.startswith(prefix) and '.LayerNorm.' in k]:
      state_dict[k.replace('.LayerNorm.', '.layer_norm.')] = state_dict.pop(k)
    state_dict.pop(prefix + 'embeddings.position_ids', None)  # a buffer of older transformers releases

  @staticmethod
  def _hf_names_out(module, state_dict, prefix, local_metadata):
    for k in [k for k in state_dict if k.startswith(prefix) and '.layer_norm.' in k]:
      state_dict[k.replace('.layer_norm.', '.LayerNorm.')] = state_dict.pop(k)
    return state_dict

  @classmethod
  def from_hf(cls, hf_model):
    """Build from an instantiated transformers BertModel (architecture + weights)."""
    m = cls(hf_model.config)
    m.load_state_dict(hf_model.state_dict(), strict=False)
    return m

  def flat_named_params(self):
    return [('embeddings.word_embeddings.weight', self.embeddings.word_embeddings.weight)] + self.engine_named_params()

  # ---- forward ----------------------------------------------------------------------------------------
  def _plan(self, bsz, seq, dev):
    key = (bsz, seq, dev)
    p = self._plans.get(key)
    if p is None:
      rows = bsz * seq
      R = ops.pad_rows(rows)
      p = types.SimpleNamespace(rows=rows, R=R, ids=torch.zeros(R, dtype=torch.int32, device=dev),
                                types=torch.zeros(R, dtype=torch.int32, device=dev),
                                pos=torch.zeros(R, dtype=torch.int32, device=dev),
                                mask_bias=torch.zeros(R, dtype=torch.float32, device=dev),
                                cls_rows=torch.arange(bsz, dtype=torch.int32, device=dev) * seq,
                                # packed variant (mmt_text_plan): one int32 block [ids | types | pos | row_index] + scalars
                                packed=torch.zeros(4, R, dtype=torch.int32, device=dev),
                                counts=torch.zeros(bsz, dtype=torch.int32, device=dev),
                                cu=torch.zeros(bsz + 1, dtype=torch.int32, device=dev),
                                n_rows=torch.zeros(1, dtype=torch.int32, device=dev),
                                packed_cls=torch.zeros(bsz, dtype=torch.int32, device=dev),
                                zero_bias=torch.zeros(R, dtype=torch.float32, device=dev))
      self._plans[key] = p
    return p

  def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
              features=None):
    if input_ids is None or features is not None:
      raise ValueError('the text tower takes input_ids (its features are the word embeddings)')
    if head_mask is not None:
      raise NotImplementedError('head_mask (the reference always passes None, model/model.py:375)')
    if not input_ids.is_cuda:
      raise RuntimeError('mmt_amd.TextBertModel runs on the GPU only (no CPU fallback)')
    bsz, seq = input_ids.shape
    dev = input_ids.device
    if seq > self.config.max_position_embeddings:
      raise ValueError('sequence length %d exceeds max_position_embeddings' % seq)
    self._ensure_ready(dev)
    p = self._plan(bsz, seq, dev)
    rows = p.rows
    table = self.embeddings.word_embeddings.weight
    if self.cls_only and self.pack_tokens and attention_mask is not None:
      # variable-length path: the engine sees only the real tokens (cu_seqlens), the [CLS] rows are read out
      def i64(x):
        return None if x is None else x.expand(bsz, seq).to(torch.int64).contiguous()
      ids64, typ64, pos64, msk64 = i64(input_ids), i64(token_type_ids), i64(position_ids), i64(attention_mask)
      p.packed.zero_()
      check(_lib.lib().mmt_text_plan(ops._p(ids64), ops._p(typ64) if typ64 is not None else None,
                                     ops._p(pos64) if pos64 is not None else None, ops._p(msk64), bsz, seq,
                                     ops._p(p.counts), ops._p(p.cu), ops._p(p.n_rows), ops._p(p.packed[0]),
                                     ops._p(p.packed[1]), ops._p(p.packed[2]), ops._p(p.packed[3]), ops._p(p.packed_cls),
                                     ops._stream()), 'mmt_text_plan')
      feats = _WordEmbeddingFn.apply(self, p.packed[0], rows, table, p.n_rows)
      batch = EngineBatch(None, p.packed[1], p.packed[2], p.zero_bias, rows, bsz, seq, cu_seqlens=p.cu,
                          row_index=p.packed[3], n_rows_dev=p.n_rows, out_rows=p.packed_cls, n_out_per_sample=1)
      batch.live_rows_hint = int(self.live_rows_hint or 0)
    else:
      p.ids[:rows].copy_(input_ids.reshape(-1))
      if token_type_ids is None:
        p.types.zero_()
      else:
        p.types[:rows].copy_(token_type_ids.reshape(-1))
      if position_ids is None:
        p.pos[:rows].copy_(torch.arange(seq, device=dev, dtype=torch.int32).repeat(bsz))
      else:
        p.pos[:rows].copy_(position_ids.expand(bsz, seq).reshape(-1))
      if attention_mask is None:
        p.mask_bias.zero_()
      else:
        torch.mul(1.0 - attention_mask.reshape(-1).to(torch.float32), -10000.0, out=p.mask_bias[:rows])
      feats = _WordEmbeddingFn.apply(self, p.ids, rows, table)
      batch = EngineBatch(None, p.types, p.pos, p.mask_bias, rows, bsz, seq,
                          out_rows=p.cls_rows if self.cls_only else None, n_out_per_sample=1 if self.cls_only else 0)
    last = self.run_engine(batch, feats)
    d = self.config.hidden_size
    if self.cls_only:
      cls = last[:bsz] if self.compact_output(batch, p.R) else last.index_select(0, batch.out_rows.long())
      seq_out = cls.view(bsz, 1, d)
    else:
      seq_out = last[:rows].view(bsz, seq, d)
    return (seq_out, self.pooler(seq_out) if self.compute_pooler and not self.cls_only else None)
