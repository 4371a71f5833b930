"""Seeded synthetic minibatches and parameter sets of MSRVTT shape.

There is no dataset in the build/bench environment, so every test, fixture and
bench line uses inputs generated here from `numpy.random.RandomState` (stable
across numpy/torch versions and machines).  The layout is exactly what the
reference's collate hands to `CENet.forward`
(data_loader/mix_dataset.py:112-144, base/base_dataset.py:572-896):

  token_ids            (B, C, W, 2) int32   [word id, valid flag]
  features[mod]        (B, T, D_mod) fp32   zero beyond the valid length
  features_t[mod]      (B, T) fp32          seconds+2 on valid tokens, 1 on padding
  features_ind[mod]    (B, T) fp32 {0,1}    first k ones; k=0 => missing expert
  features_avgpool/maxpool[mod] (B, D_mod)  pooled over the valid rows (0 if none)
  query_masks          (B, C) fp32
"""
import collections
import zlib

import numpy as np
import torch

# utils/util.py:154-247 of the reference: expert -> (feature dim, token-type idx)
EXPERT_TABLE = {
    's3d': (1024, 1), 'vggish': (128, 2), 'face': (512, 3), 'audio': (128, 4),
    'rgb': (2048, 5), 'speech': (300, 6), 'ocr': (300, 7), 'flow': (1024, 8),
    'scene': (2208, 9),
}
MSRVTT_MODALITIES = ['face', 'ocr', 'rgb', 's3d', 'scene', 'speech', 'vggish']


def compute_dims(modalities, face_dim=512):
  dims = collections.OrderedDict()
  for mod in sorted(modalities):
    dim, idx = EXPERT_TABLE[mod]
    dims[mod] = {'dim': face_dim if mod == 'face' else dim, 'idx': idx}
  return dims


def vid_bert_params(hidden=512, layers=4, heads=4, inter=3072, max_pos=32, dropout=0.1):
  """configs_pub/eccv20/MSRVTT_jsfusion_trainval.json:30-43."""
  return {
      'vocab_size_or_config_json_file': 10, 'hidden_size': hidden,
      'num_hidden_layers': layers, 'num_attention_heads': heads,
      'intermediate_size': inter, 'hidden_act': 'gelu',
      'hidden_dropout_prob': dropout, 'attention_probs_dropout_prob': dropout,
      'max_position_embeddings': max_pos, 'type_vocab_size': 19,
      'initializer_range': 0.02, 'layer_norm_eps': 1e-12,
  }


def make_batch(seed, batch, modalities, max_tokens=30, captions=1, max_words=30,
               max_pos=32, text_dim=768, missing_prob=None, fill=None):
  """Returns (minibatch dict of CPU tensors, text (B,C,text_dim) fp32).

  `text` stands in for the output of the (out-of-scope) text tower.  fill (None = SURVEY 8d's U{0..max_tokens} valid
  lengths, mean fill 0.5): mean fraction of the feature-token slots that hold a real token -- valid lengths uniform on
  {0..2 fill max_tokens} (fill <= 0.5) or on {max_tokens (2 fill - 1)..max_tokens} (fill > 0.5); bench.py --fill."""
  rs = np.random.RandomState(seed)
  dims = compute_dims(modalities)
  mb = {k: collections.OrderedDict() for k in
        ('features', 'features_t', 'features_ind', 'features_avgpool', 'features_maxpool')}
  for mod in dims:
    d = dims[mod]['dim']
    k = rs.randint(0, max_tokens + 1, size=batch)  # valid length, 0 => missing
    if fill is not None:  # (same draws, rescaled: the default stream of the golden fixtures is untouched)
      f = min(max(float(fill), 0.0), 1.0)
      if f <= 0.5:
        k = np.round(k * (2.0 * f)).astype(k.dtype)
      else:
        k = (max_tokens - np.round((max_tokens - k) * (2.0 - 2.0 * f))).astype(k.dtype)
    if missing_prob is not None:
      k = np.where(rs.rand(batch) < missing_prob, 0, np.maximum(k, 1))
    ind = (np.arange(max_tokens)[None, :] < k[:, None]).astype(np.float32)
    feats = rs.randn(batch, max_tokens, d).astype(np.float32) * ind[:, :, None]
    t = np.sort(rs.uniform(2, max_pos - 2, size=(batch, max_tokens)), axis=1).astype(np.float32)
    t = np.where(ind > 0, t, 1.0).astype(np.float32)
    denom = np.maximum(k, 1)[:, None].astype(np.float32)
    avg = feats.sum(1) / denom
    mx = np.where(ind[:, :, None] > 0, feats, -np.inf).max(1)
    mx = np.where(np.isfinite(mx), mx, 0.0).astype(np.float32)
    mb['features'][mod] = torch.from_numpy(feats)
    mb['features_t'][mod] = torch.from_numpy(t)
    mb['features_ind'][mod] = torch.from_numpy(ind)
    mb['features_avgpool'][mod] = torch.from_numpy(avg.astype(np.float32))
    mb['features_maxpool'][mod] = torch.from_numpy(mx)
  ids = rs.randint(1000, 20000, size=(batch, captions, max_words)).astype(np.int32)
  n = rs.randint(5, max_words + 1, size=(batch, captions))
  valid = (np.arange(max_words)[None, None, :] < n[:, :, None]).astype(np.int32)
  mb['token_ids'] = torch.from_numpy(np.stack([ids * valid, valid], -1))
  mb['query_masks'] = torch.ones(batch, captions)
  text = torch.from_numpy(rs.randn(batch, captions, text_dim).astype(np.float32))
  return mb, text


def _rs_for(seed, name):
  return np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))


def make_param(seed, name, shape):
  """Deterministic fp32 value for one state_dict entry, by name.

  Scales are chosen so that every term of the path is exercised (non-zero
  biases, non-trivial LayerNorm/BatchNorm affine, running stats != identity)."""
  rs = _rs_for(seed, name)
  leaf = name.rsplit('.', 1)[-1]
  if leaf == 'num_batches_tracked':
    return torch.tensor(3, dtype=torch.long)
  if leaf == 'running_var':
    return torch.from_numpy(rs.uniform(0.5, 1.5, size=shape).astype(np.float32))
  if leaf == 'running_mean':
    return torch.from_numpy((0.1 * rs.randn(*shape)).astype(np.float32))
  norm_like = 'layer_norm' in name or 'batch_norm' in name
  if leaf == 'weight' and norm_like:
    return torch.from_numpy((1.0 + 0.1 * rs.randn(*shape)).astype(np.float32))
  if leaf == 'bias':
    scale = 0.1 if norm_like else 0.02
    return torch.from_numpy((scale * rs.randn(*shape)).astype(np.float32))
  if len(shape) == 2:
    if name.startswith('vid_bert.') or name.startswith('embeddings.') or name.startswith('encoder.'):
      std = 0.05  # a little above initializer_range so that attention is not uniform
    else:
      std = 1.0 / np.sqrt(shape[1])
    return torch.from_numpy((std * rs.randn(*shape)).astype(np.float32))
  return torch.from_numpy((0.02 * rs.randn(*shape)).astype(np.float32))


def make_state_dict(seed, shapes):
  """shapes: {name: tuple}.  Returns an OrderedDict of fp32 CPU tensors."""
  return collections.OrderedDict((n, make_param(seed, n, tuple(s))) for n, s in shapes.items())


def checksum(t):
  """Order-sensitive fp64 checksum used by fixtures to detect generator drift."""
  a = t.detach().double().reshape(-1).numpy()
  w = np.cos(np.arange(a.size, dtype=np.float64) * 0.37) + 1.5
  return float((a * w).sum())


def text_token_batch(seed, b, w, vocab):
  """(input_ids, attention_mask) int64 (b, w) like the reference's token_ids[..., 0] / [..., 1]
  (base/base_dataset.py:874-896): [CLS]-led captions of 5..w tokens, zero-padded tails, with tokens repeated inside a
  caption and across captions (the word-embedding gradient has to accumulate them)."""
  rs = np.random.RandomState(seed)
  ids = rs.randint(1, vocab, size=(b, w)).astype(np.int64)
  ids[:, 0] = 101 % vocab
  ids[:, 3] = ids[:, 1]
  ids[1:, 2] = ids[0, 2]
  lens = rs.randint(5, w + 1, size=b)
  lens[0] = w
  mask = (np.arange(w)[None, :] < lens[:, None]).astype(np.int64)
  ids = ids * mask
  return torch.from_numpy(ids), torch.from_numpy(mask)
