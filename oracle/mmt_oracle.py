"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY, never part of the product path.

A functional restatement (plain torch on CPU, any float dtype, autograd gives the
gradients) of the MMT hot path named by BASELINE.json:north_star.  Every
function cites the reference file:line it follows (paths relative to
/root/reference).  Parameters are passed as a flat dict keyed by the
reference's own state_dict names so that a reference checkpoint can be fed in
unchanged.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module.  Nothing under `mmt_amd/` does: the product
path has no CPU fallback and fails loudly without the HIP library.

PINNING: the reference ships no tests/golden vectors for this path (SURVEY.md
section 4), so the oracle is pinned against the reference *itself*, executed on
CPU in the build container: `oracle/gen_golden.py` imports /root/reference,
runs reference and oracle on identical seeded inputs, asserts agreement and
writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` re-checks the oracle
against those fixtures wherever the reference tree is absent (the GPU box).

Dropout: the reference draws ATen Philox masks which cannot be reproduced; the
oracle takes the keep-masks as *inputs* (`masks[name]`, float 0/1) so that the
native path's own masks can be replayed through it.  masks=None means p=0/eval.
"""
import collections
import math

import numpy as np
import torch
import torch.nn.functional as F

# utils/util.py:154-247 -- expert -> (input dim, token-type idx); sorted() order.
EXPERT_TABLE = {
    's3d': (1024, 1), 'vggish': (128, 2), 'face': (512, 3), 'audio': (128, 4),
    'rgb': (2048, 5), 'speech': (300, 6), 'ocr': (300, 7), 'flow': (1024, 8),
    'scene': (2208, 9),
}


def compute_dims(modalities, face_dim=512):
  """utils/util.py:154-247 (the experts used by the eccv20 configs)."""
  dims = collections.OrderedDict()
  for mod in sorted(modalities):
    dim, idx = EXPERT_TABLE[mod]
    if mod == 'face':
      dim = face_dim
    dims[mod] = {'dim': dim, 'idx': idx}
  return dims


# ----------------------------------------------------------------------------
# model/bert.py
# ----------------------------------------------------------------------------
def gelu_erf(x):
  """model/bert.py:37-53 -- erf form, not the tanh approximation."""
  return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, weight, bias, eps):
  """torch.nn.LayerNorm as aliased at model/bert.py:71 (biased variance)."""
  mu = x.mean(-1, keepdim=True)
  var = ((x - mu) ** 2).mean(-1, keepdim=True)
  return (x - mu) / torch.sqrt(var + eps) * weight + bias


def _drop(x, masks, name, p):
  if masks is None or p == 0.0:
    return x
  return x * masks[name].to(x.dtype) / (1.0 - p)


def bert_embeddings(P, pre, token_type_ids, position_ids, features, eps, masks=None, p=0.0):
  """model/bert.py:87-105."""
  emb = P[pre + 'token_type_embeddings.weight'][token_type_ids] + features
  if position_ids is not None:
    emb = emb + P[pre + 'position_embeddings.weight'][position_ids]
  emb = layer_norm(emb, P[pre + 'layer_norm.weight'], P[pre + 'layer_norm.bias'], eps)
  return _drop(emb, masks, 'emb', p)


def bert_self_attention(P, pre, h, ext_mask, num_heads, masks=None, p=0.0, tag=''):
  """model/bert.py:136-172."""
  b, s, d = h.shape
  dh = d // num_heads

  def split(x):
    return x.view(b, s, num_heads, dh).permute(0, 2, 1, 3)

  q = split(F.linear(h, P[pre + 'query.weight'], P[pre + 'query.bias']))
  k = split(F.linear(h, P[pre + 'key.weight'], P[pre + 'key.bias']))
  v = split(F.linear(h, P[pre + 'value.weight'], P[pre + 'value.bias']))
  scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh) + ext_mask
  probs = torch.softmax(scores, dim=-1)
  probs = _drop(probs, masks, tag + 'probs', p)
  ctx = torch.matmul(probs, v)
  return ctx.permute(0, 2, 1, 3).contiguous().view(b, s, d)


def bert_layer(P, pre, h, ext_mask, cfg, masks=None, tag=''):
  """model/bert.py:240-256 (= :136-172 + :185-189 + :217-220 + :233-237)."""
  eps = cfg['layer_norm_eps']
  ph = cfg['hidden_dropout_prob'] if masks is not None else 0.0
  pa = cfg['attention_probs_dropout_prob'] if masks is not None else 0.0
  ctx = bert_self_attention(P, pre + 'attention.self.', h, ext_mask,
                            cfg['num_attention_heads'], masks, pa, tag)
  a = F.linear(ctx, P[pre + 'attention.output.dense.weight'], P[pre + 'attention.output.dense.bias'])
  a = layer_norm(_drop(a, masks, tag + 'attn_out', ph) + h,
                 P[pre + 'attention.output.layer_norm.weight'],
                 P[pre + 'attention.output.layer_norm.bias'], eps)
  i = gelu_erf(F.linear(a, P[pre + 'intermediate.dense.weight'], P[pre + 'intermediate.dense.bias']))
  o = F.linear(i, P[pre + 'output.dense.weight'], P[pre + 'output.dense.bias'])
  return layer_norm(_drop(o, masks, tag + 'ffn_out', ph) + a,
                    P[pre + 'output.layer_norm.weight'], P[pre + 'output.layer_norm.bias'], eps)


def bert_model(P, pre, cfg, attention_mask, token_type_ids, position_ids, features,
               masks=None, with_pooler=False):
  """model/bert.py:371-414.  Returns sequence_output (and pooled_output)."""
  dtype = features.dtype
  ext = (1.0 - attention_mask[:, None, None, :].to(dtype)) * -10000.0  # :386-395
  ph = cfg['hidden_dropout_prob'] if masks is not None else 0.0
  h = bert_embeddings(P, pre + 'embeddings.', token_type_ids, position_ids, features,
                      cfg['layer_norm_eps'], masks, ph)
  for l in range(cfg['num_hidden_layers']):
    h = bert_layer(P, pre + 'encoder.layer.%d.' % l, h, ext, cfg, masks, 'l%d.' % l)
  if not with_pooler:
    return h
  pooled = torch.tanh(F.linear(h[:, 0], P[pre + 'pooler.dense.weight'], P[pre + 'pooler.dense.bias']))
  return h, pooled  # :306-308


def text_bert_model(P, pre, cfg, input_ids, attention_mask, token_type_ids=None, position_ids=None):
  """The text tower: HuggingFace `BertModel` as the reference instantiates and calls it (model/model.py:152-162,
  371-376; third-party transformers==3.1.0, requirements.txt:42, modeling_bert.py BertEmbeddings/BertEncoder):
  word + position + token-type embeddings -> LayerNorm -> dropout -> the same encoder layers as model/bert.py with
  head dim hidden/heads.  `P` uses HuggingFace key names (`LayerNorm`).  Returns sequence_output (B, W, hidden).
  Pinned against the installed transformers' BertModel by oracle/gen_golden.py (tests/golden/text_bert.npz)."""
  Q = {k.replace('.LayerNorm.', '.layer_norm.'): v for k, v in P.items()}
  b, w = input_ids.shape
  if position_ids is None:
    position_ids = torch.arange(w).unsqueeze(0).expand(b, w)
  if token_type_ids is None:
    token_type_ids = torch.zeros_like(input_ids)
  features = Q[pre + 'embeddings.word_embeddings.weight'][input_ids]
  return bert_model(Q, pre, cfg, attention_mask, token_type_ids, position_ids, features)


# ----------------------------------------------------------------------------
# model/model.py
# ----------------------------------------------------------------------------
def l2_normalize(x, eps=1e-12):
  """F.normalize(x, dim=-1): x / max(||x||_2, eps)."""
  return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def reduce_dim(P, pre, x):
  """model/model.py:717-726."""
  return l2_normalize(F.linear(x, P[pre + 'fc.weight'], P[pre + 'fc.bias']))


def gated_embedding_unit(P, pre, x, training, bn_eps=1e-5):
  """model/model.py:683-702 + ContextGating :736-750 (use_bn=True, normalize=True)."""
  x = F.linear(x, P[pre + 'fc.weight'], P[pre + 'fc.bias'])
  x1 = F.linear(x, P[pre + 'cg.fc.weight'], P[pre + 'cg.fc.bias'])
  if training:
    mu = x1.mean(0, keepdim=True)
    var = ((x1 - mu) ** 2).mean(0, keepdim=True)
  else:
    mu = P[pre + 'cg.batch_norm.running_mean'][None]
    var = P[pre + 'cg.batch_norm.running_var'][None]
  x1 = (x1 - mu) / torch.sqrt(var + bn_eps) * P[pre + 'cg.batch_norm.weight'] + P[pre + 'cg.batch_norm.bias']
  return l2_normalize(x * torch.sigmoid(x1))  # F.glu(cat(x, x1), 1)


def video_token_layout(modalities, max_tokens):
  """Index bookkeeping of model/model.py:485-567: token 0 = CLS, then per expert
  one AGG token followed by T FEA tokens.  Returns (S, {mod: agg_index})."""
  tok = 0
  agg = collections.OrderedDict()
  for mod in modalities:
    tok += 1
    agg[mod] = tok
    tok += max_tokens[mod]
  return tok + 1, agg


def assemble_video_tokens(P, modalities, expert_dims, batch, same_dim, max_pos):
  """model/model.py:426-437 (ReduceDim) + :485-567 (token assembly) for
  vid_inp='both', out_tok='mxp', pos_enc='tint'.  `batch` holds features,
  features_t, features_ind, features_maxpool dicts.  Returns features (B,S,d),
  token_type_ids, position_ids, attention_mask (B,S) int64 and the AGG map."""
  b = batch['features'][modalities[0]].shape[0]
  dtype = batch['features'][modalities[0]].dtype
  feats, types, poss, masks = [], [], [], []
  feats.append(torch.zeros(b, 1, same_dim, dtype=dtype))  # CLS :497-504
  types.append(torch.zeros(b, 1, dtype=torch.long))
  poss.append(torch.zeros(b, 1, dtype=torch.long))
  masks.append(torch.ones(b, 1, dtype=torch.long))
  max_tokens = collections.OrderedDict()
  for mod in modalities:
    pre = 'video_dim_reduce.%s.' % mod
    t = batch['features'][mod].shape[1]
    max_tokens[mod] = t
    tt = expert_dims[mod]['idx']
    ind = batch['features_ind'][mod]
    # AGG token :521-541
    feats.append(reduce_dim(P, pre, batch['features_maxpool'][mod])[:, None])
    types.append(torch.full((b, 1), tt, dtype=torch.long))
    poss.append(torch.zeros(b, 1, dtype=torch.long))
    masks.append(ind.max(1)[0].long()[:, None])  # :330, :541
    # FEA tokens :542-558
    feats.append(reduce_dim(P, pre, batch['features'][mod]))
    types.append(torch.full((b, t), tt, dtype=torch.long))
    poss.append(batch['features_t'][mod].clamp(0, max_pos).long())  # :513-520
    masks.append(ind.long())
  s, agg = video_token_layout(modalities, max_tokens)
  return (torch.cat(feats, 1), torch.cat(types, 1), torch.cat(poss, 1),
          torch.cat(masks, 1), agg)


def text_moe_weights(P, modalities, text, masks=None, p=0.0):
  """model/model.py:262-283 (text branch) + :618.  text: (B, C, 768) -> (B, C, M).
  masks['moe'] (same shape as text) replays moe_txt_dropout (:274) in train mode."""
  if masks is not None and 'moe' in masks and p > 0.0:
    text = text * masks['moe'].to(text.dtype).view_as(text) / (1.0 - p)
  logits = torch.cat([F.linear(text, P['moe_fc_txt.%s.weight' % m], P['moe_fc_txt.%s.bias' % m])
                      for m in modalities], dim=-1)
  w = torch.softmax(logits, dim=-1)
  return w / w.abs().sum(-1, keepdim=True).clamp_min(1e-12)  # F.normalize(p=1)


def cross_view_inner_product(vid_embds, text_embds, vid_weights, text_weights, merge='avg'):
  """model/model.py:789-837 in tensor form.
  vid_embds (B,M,d); text_embds (B,M,C,d); vid_weights (B,M); text_weights (B,C,M).
  Returns (B*C, B) ['indep' or C==1] or (B, B) ['avg']; rows = text, cols = video."""
  b, m, d = vid_embds.shape
  c = text_embds.shape[2]
  tw = text_weights.reshape(b * c, m)
  moe = tw[:, None, :] * vid_weights[None, :, :]          # :810
  norm = moe.sum(-1, keepdim=True)
  norm = torch.where(norm == 0, torch.full_like(norm, 1e-5), norm)  # :816
  moe = moe / norm
  te = text_embds.permute(0, 2, 1, 3).reshape(b * c, m, d)  # row = b*C + cap :805,822
  sims = torch.einsum('tvm,tmd,vmd->tv', moe, te, vid_embds)
  if c > 1:
    if merge == 'avg':
      sims = sims.view(b, c, b).mean(1)
    elif merge != 'indep':
      raise ValueError('unrecognised merge mode: {}'.format(merge))
  return sims


def max_margin_ranking_loss(x, margin=0.05, fix_norm=True):
  """model/loss.py:38-65 in closed form: both hinge directions over the
  (off-diagonal if fix_norm) entries, mean over 2n(n-1) [or 2n^2]."""
  n = x.shape[0]
  diag = torch.diagonal(x)
  h = torch.relu(margin - diag[:, None] + x) + torch.relu(margin - diag[:, None] + x.t())
  if fix_norm:
    h = h * (1.0 - torch.eye(n, dtype=x.dtype))
    return h.sum() / (2.0 * n * (n - 1))
  return h.sum() / (2.0 * n * n)


def cross_view_rows(txt_rows, tw_rows, vid_all, vw_all):
  """model/model.py:789-837 for a SUBSET of the text rows against every video (one caption per video):
  txt_rows (r, M, d), tw_rows (r, M), vid_all (n, M, d), vw_all (n, M) -> (r, n).  Same arithmetic as
  cross_view_inner_product, without the square-batch assumption -- for sampled-row checks of matrices too large to form."""
  moe = tw_rows[:, None, :] * vid_all.new_tensor(1.0) * vw_all[None, :, :]                     # :810
  norm = moe.sum(-1, keepdim=True)
  norm = torch.where(norm == 0, torch.full_like(norm, 1e-5), norm)                            # :816
  dots = torch.stack([txt_rows[:, m] @ vid_all[:, m].t() for m in range(txt_rows.shape[1])], -1)   # (r, n, M)
  return ((moe / norm) * dots).sum(-1)


def max_margin_rows(s_rows, row_ids, diag_all, margin, n_total, col_hinge_counts=None, fix_norm=True):
  """model/loss.py:38-65 restricted to the text rows `row_ids` of an n x n similarity matrix (fix_norm=True):
  entry (R, c), c != R, contributes relu(m - s_RR + s_Rc) [row R's hinge] + relu(m - s_cc + s_Rc) [column c's hinge].
  s_rows (r, n); diag_all (n) = s_jj.  -> (loss contribution of these rows, d loss / d s_rows) with the diagonal entry's
  gradient -(#active row hinges of R + col_hinge_counts[i]) / norm, col_hinge_counts[i] = #active column hinges
  relu(m - s_RR + s_r'R) over the OTHER rows r' (they live outside s_rows; 0 if not given)."""
  assert fix_norm
  r, n = s_rows.shape
  norm = 2.0 * n_total * (n_total - 1)
  off = torch.ones_like(s_rows)
  off[torch.arange(r), row_ids] = 0.0
  a_row = (margin - diag_all[row_ids][:, None] + s_rows) * off
  a_col = (margin - diag_all[None, :] + s_rows) * off
  loss = (torch.relu(a_row) + torch.relu(a_col)).sum(1) / norm
  g = ((a_row > 0).to(s_rows.dtype) + (a_col > 0).to(s_rows.dtype)) * off / norm
  rowcnt = (a_row > 0).sum(1)
  cc = torch.zeros_like(rowcnt) if col_hinge_counts is None else col_hinge_counts
  g[torch.arange(r), row_ids] = -(rowcnt + cc).to(s_rows.dtype) / norm
  return loss, g, rowcnt


def info_nce_loss(x):
  """model/loss.py:68-81."""
  t = torch.arange(x.shape[0])
  return F.cross_entropy(x, t) + F.cross_entropy(x.t(), t)


def cenet_forward(P, cfg, batch, text, training, masks=None, out='conf'):
  """model/model.py:312-661 for the configuration every published config uses
  (vid_cont='bert', vid_inp='both', out_tok='mxp', pos_enc='tint', vid_wgh='none',
  txt_wgh='emb', txt_pro='gbn', keep_missing_modalities=True, normalize_experts).
  `text` = (B, C, 768) output of the text tower (out of scope).
  cfg: {'modalities', 'expert_dims', 'vid_bert_params', 'same_dim', 'test_caption_mode'}."""
  mods = cfg['modalities']
  vb = cfg['vid_bert_params']
  b, c, _ = text.shape
  text_embd = [gated_embedding_unit(P, 'text_GU.%s.' % m, text.reshape(b * c, -1), training).view(b, c, -1)
               for m in mods]  # :413-417
  feats, types, poss, amask, agg = assemble_video_tokens(
      P, mods, cfg['expert_dims'], batch, cfg['same_dim'], vb['max_position_embeddings'] - 1)
  last = bert_model(P, 'vid_bert.', vb, amask, types, poss, feats, masks)  # :577-581
  vid = torch.stack([l2_normalize(last[:, agg[m]]) for m in mods], 1)  # :583-587, :621-625
  txt = torch.stack([l2_normalize(t) for t in text_embd], 1)  # (B,M,C,d) second normalise :623
  vw = torch.full((b, len(mods)), 1.0 / len(mods), dtype=text.dtype)  # :594,607
  # :610-618; moe_txt_dropout (:274) only when its keep-mask is supplied (cfg['moe_dropout_prob'], masks['moe'])
  tw = text_moe_weights(P, mods, text, masks, float(cfg.get('moe_dropout_prob', 0.0)))
  if out != 'conf':
    return {'vid_embds': vid, 'text_embds': txt, 'vid_weights': vw, 'text_weights': tw}
  merge = 'avg' if training else cfg.get('test_caption_mode', 'indep')  # :627-631
  return {'modalities': mods,
          'cross_view_conf_matrix': cross_view_inner_product(vid, txt, vw, tw, merge)}


# ----------------------------------------------------------------------------
# model/metric.py (restated as counting ranks; tie-averaging as :90-121)
# ----------------------------------------------------------------------------
def _cols2metrics(cols, nq):
  """model/metric.py:246-258."""
  cols = np.asarray(cols, dtype=np.float64)
  out = {'R1': 100 * float(np.sum(cols == 0)) / nq, 'R5': 100 * float(np.sum(cols < 5)) / nq,
         'R10': 100 * float(np.sum(cols < 10)) / nq, 'R50': 100 * float(np.sum(cols < 50)) / nq,
         'MedR': float(np.median(cols) + 1), 'MeanR': float(np.mean(cols) + 1)}
  stats = np.array([out['R1'], out['R5'], out['R10']])
  out['geometric_mean_R1-R5-R10'] = float(np.exp(np.mean(np.log(stats)))) if (stats > 0).all() else 0.0
  return out


def t2v_metrics(sims, query_masks=None):
  """model/metric.py:26-150: rank of the ground-truth video per text query,
  ties averaged = #strictly-closer + (#equal-1)/2."""
  sims = np.asarray(sims)
  nq, nv = sims.shape
  qu = nq // nv
  d = -sims
  gt = d[np.arange(nq), np.arange(nq) // qu][:, None]
  cols = (d < gt).sum(1) + ((d == gt).sum(1) - 1) / 2.0
  if query_masks is not None:
    keep = np.asarray(query_masks).reshape(-1).astype(bool)
    cols, nq = cols[keep], int(keep.sum())
  return _cols2metrics(cols, nq)


def v2t_metrics(sims, query_masks=None):
  """model/metric.py:153-243: per video the best (min) tie-averaged rank among
  its own captions; missing captions sit at distance 1e8."""
  d = -np.asarray(sims).T.copy()
  nv, nc = d.shape
  cpv = nc // nv
  if query_masks is not None:
    d[:, np.logical_not(np.asarray(query_masks).reshape(-1).astype(bool))] = 1e8
  ranks = []
  for i in range(nv):
    row = d[i]
    best = np.inf
    for j in range(i * cpv, (i + 1) * cpv):
      if row[j] == 1e8:
        continue
      r = (row < row[j]).sum() + ((row == row[j]).sum() - 1) / 2.0
      best = min(best, r)
    ranks.append(best)
  return _cols2metrics(ranks, nv)
