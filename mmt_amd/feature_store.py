"""Input side of the hot path: memory-mapped feature shards -> per-sample selection ->
ragged bf16 wire buffer -> device.  What it stands in for on the reference side:

  base/base_dataset.py:426-497        get_sample_data: one h5 file PER VIDEO, opened and parsed at every __getitem__
  base/base_dataset.py:357-379        get_feature_timings (utils/expert_timings.py holds the per-expert widths)
  base/base_dataset.py:71-113         choose_or_pad_to_len: pick <= max_tokens rows, zero-pad to max_tokens
  base/base_dataset.py:765-850        per-expert window selection, max pooling, padding of missing experts
  data_loader/mix_dataset.py:112-144  collate_data: dict of dense fp32 (B, T, D_expert) arrays
  trainer/trainer.py:36-52            move_dict_to_device: one blocking upload per tensor (~35 of them)

MI355X-first layout.  The device consumes the features of expert e as ONE compact bf16 matrix X_e (csrc/assemble.hip):
rows [0, B) = the max-pooled vector of each sample, rows B + i = the VALID feature rows in (sample, time) order, K
zero-padded to a multiple of 128.  The wire format IS that matrix: the collator gathers rows from the memory-mapped store
straight into a pinned buffer with the device layout, the upload copies the live prefix of every X_e (M + 1 asynchronous
copies) and the ReduceDim GEMM reads it in place -- no dense (B, T, D) fp32 tensor, no padding rows over PCIe (4x fewer
bytes at the synthetic-MSRVTT fill: bf16 x ~52 % live rows), no cast kernel.  bf16 on the wire changes nothing: the
device path rounds the fp32 features to bf16 (round-to-nearest-even) before the first GEMM anyway, `to_bf16` below is the
same rounding, and max pooling commutes with a monotone rounding.

Not carried: features_avgpool (only read for out_tok='avg', which the drop-in rejects).  Captions are stored as the token
ids the (HuggingFace, upstream) tokenizer produced for `caption_text(words)`; `collate_tokens` applies the cut to
max_text_words / forced [SEP] / padding rules of the reference (base_dataset.py:320-346, 63-68)."""
import json
import os

import numpy as np
import torch

__all__ = ['FEAT_WIDTH', 'feature_timings', 'choose_rows', 'caption_text', 'crop_or_pad_tokens', 'to_bf16', 'from_bf16', 'FeatureStoreWriter', 'FeatureStore',
           'RaggedLayout', 'RaggedFeatures', 'RaggedCollator']

# utils/expert_timings.py: seconds covered by one feature row (stride = width); every other expert has no timing (-1)
FEAT_WIDTH = {'rgb': 0.2, 'scene': 1.0, 's3d': 1.0, 'vggish': 1.0}
# base/base_dataset.py:476: only these experts' stored timings are trusted, the others are recomputed from FEAT_WIDTH
STORED_TIMING_EXPERTS = ('s3d', 'vggish')


def _round_up(x, m):
  return (x + m - 1) // m * m


def feature_timings(nb_feats, feat_width, stride=None, group=None):
  """[nb_feats, 2] start/end second of every feature row (base/base_dataset.py:357-379)."""
  if feat_width is None:
    return np.full((nb_feats, 2), -1.0)
  if group is not None:
    if nb_feats % group:
      raise ValueError('nb_feats must be a multiple of group')
    return np.repeat(feature_timings(nb_feats // group, feat_width, stride), group, axis=-1)
  stride = feat_width if stride is None else stride
  last = (nb_feats - 1) * stride
  return np.stack((np.linspace(0, last, num=nb_feats), np.linspace(feat_width, last + feat_width, num=nb_feats)), axis=-1)


def choose_rows(n, max_tokens, training, rng=None):
  """Sorted indices of the rows `choose_or_pad_to_len` keeps (base/base_dataset.py:96-104): min(n, max_tokens) of n
  without replacement; training draws from `rng` (np.random, the reference's global generator, when None), evaluation
  from a fresh RandomState(0) so every epoch sees the same rows."""
  keep = min(n, max_tokens)
  if training:
    pick = (np.random if rng is None else rng).choice(n, size=keep, replace=False)
  else:
    pick = np.random.RandomState(0).choice(n, size=keep, replace=False)
  return np.sort(pick)


def caption_text(words):
  """The string `tokenize_caption` hands to the tokenizer (base_dataset.py:328-334): words joined by blanks, stripped,
  a period appended unless it already ends a sentence, capitalised."""
  txt = ' '.join(words).strip()
  if txt[-1] not in ['.', '?', '!']:
    txt += '.'
  return txt.capitalize()


def crop_or_pad_tokens(ids, max_text_words, sep_id=None):
  """[max_text_words, 2] (id, valid) rows of one caption: the token list is cut to max_text_words and, when special
  tokens are in use, its last kept token forced to [SEP] (base_dataset.py:340-344), then `crop_or_pad_to_len` (:63-68)."""
  ids = np.array(ids[:max_text_words], dtype=np.int64)
  if sep_id is not None and ids.shape[0]:
    ids[-1] = sep_id
  out = np.zeros((max_text_words, 2))
  out[:ids.shape[0], 0] = ids
  out[:ids.shape[0], 1] = 1
  return out


def to_bf16(a):
  """fp32 array -> uint16 bf16 bit patterns, round-to-nearest-even (what csrc pack_bf2 does on the device)."""
  u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
  return ((u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) >> np.uint32(16)).astype(np.uint16)


def from_bf16(u):
  return (np.asarray(u, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


# ---- the store -------------------------------------------------------------------------------------------------------
class FeatureStoreWriter:
  """One directory = one shard set: per expert `<e>.rows` ([total_rows, D] bf16 or fp32), `<e>.t` ([total_rows] f64, the
  mean of the start and end second of the row: base_dataset.py:493), `<e>.off` ([videos + 1] int64), plus meta.json.
  An expert with no usable features for a video (absent, empty, or NaN in its first value: base_dataset.py:469-472)
  contributes zero rows."""

  def __init__(self, path, experts, dtype='bf16'):
    if dtype not in ('bf16', 'f32'):
      raise ValueError('dtype must be bf16 or f32')
    os.makedirs(path, exist_ok=True)
    self.path, self.experts, self.dtype = path, dict(experts), dtype
    self.videos = []
    self._rows = {e: open(os.path.join(path, e + '.rows'), 'wb') for e in self.experts}
    self._t = {e: open(os.path.join(path, e + '.t'), 'wb') for e in self.experts}
    self._off = {e: [0] for e in self.experts}
    self._tok = open(os.path.join(path, 'captions.tok'), 'wb')
    self._cap_off, self._vid_cap = [0], [0]   # token offset of every caption, caption offset of every video
    # word-level captions (for the caption sampling modes that cut / shuffle / concatenate WORDS before tokenising):
    # UTF-8 words separated by \x1f, [start, end] seconds per word
    self._words = open(os.path.join(path, 'captions.words'), 'wb')
    self._wt = open(os.path.join(path, 'captions.wt'), 'wb')
    self._wbyte, self._wcount, self._vid_wcap = [0], [0], [0]  # byte / word offset of every caption, caption offset per video

  def add(self, vid, features, features_t=None, captions=None, caption_words=None, caption_times=None):
    """features: {expert: [n, D] float array}; features_t: {expert: [n, 2] start/end seconds} (optional per expert);
    captions: list of token-id sequences, one per caption of the video, tokenised as `tokenize_caption` does
    (base_dataset.py:320-346: `caption_text(words)` through the tokenizer, [CLS] ... [SEP], NOT yet cut to max_text_words).
    caption_words / caption_times: the captions as the reference's files hold them (`raw_captions.<i>`: word strings,
    `raw_captions_t.<i>`: [n_words, 2] seconds, base_dataset.py:446-458; times default to zeros, surplus rows are cut) --
    what `RaggedCollator.collate_captions` samples from."""
    features_t = features_t or {}
    for c, words in enumerate(caption_words or []):
      words = [w if isinstance(w, str) else w.decode('UTF-8') for w in words]
      if any('\x1f' in w for w in words):
        raise ValueError('caption words must not contain the unit separator')
      t = caption_times[c] if caption_times is not None and caption_times[c] is not None else np.zeros((len(words), 2))
      t = np.asarray(t, dtype=np.float64).reshape(-1, 2)[:len(words)]
      if t.shape[0] != len(words):
        raise ValueError('fewer caption timings than words')
      blob = '\x1f'.join(words).encode('UTF-8')
      self._words.write(blob)
      self._wt.write(np.ascontiguousarray(t).tobytes())
      self._wbyte.append(self._wbyte[-1] + len(blob))
      self._wcount.append(self._wcount[-1] + len(words))
    self._vid_wcap.append(len(self._wcount) - 1)
    for ids in (captions or []):
      ids = np.asarray(ids, dtype=np.int32).reshape(-1)
      self._tok.write(ids.tobytes())
      self._cap_off.append(self._cap_off[-1] + ids.shape[0])
    self._vid_cap.append(len(self._cap_off) - 1)
    for e, dim in self.experts.items():
      x = features.get(e)
      n = 0
      if x is not None and len(x) > 0 and not np.isnan(x[0][0]):
        x = np.asarray(x)
        if x.ndim != 2 or x.shape[1] != dim:
          raise ValueError('%s: expected [n, %d] features, got %s' % (e, dim, x.shape))
        n = x.shape[0]
        t = features_t.get(e) if e in STORED_TIMING_EXPERTS else None
        if t is not None:
          t = np.asarray(t, dtype=np.float64)[:n]  # base_dataset.py:480-484 (surplus timings are cut)
          if t.shape[0] != n:
            raise ValueError('%s: fewer timings than feature rows' % e)
        else:
          t = feature_timings(n, FEAT_WIDTH.get(e))
        self._t[e].write(np.average(t, axis=1).astype(np.float64).tobytes())
        self._rows[e].write((to_bf16(x) if self.dtype == 'bf16' else np.ascontiguousarray(x, np.float32)).tobytes())
      self._off[e].append(self._off[e][-1] + n)
    self.videos.append(str(vid))

  def close(self):
    for e in self.experts:
      self._rows[e].close()
      self._t[e].close()
      np.asarray(self._off[e], dtype=np.int64).tofile(os.path.join(self.path, e + '.off'))
    self._tok.close()
    np.asarray(self._cap_off, dtype=np.int64).tofile(os.path.join(self.path, 'captions.off'))
    np.asarray(self._vid_cap, dtype=np.int64).tofile(os.path.join(self.path, 'captions.vid'))
    self._words.close()
    self._wt.close()
    np.asarray([self._wbyte, self._wcount], dtype=np.int64).tofile(os.path.join(self.path, 'captions.woff'))
    np.asarray(self._vid_wcap, dtype=np.int64).tofile(os.path.join(self.path, 'captions.wvid'))
    with open(os.path.join(self.path, 'meta.json'), 'w') as f:
      json.dump({'version': 1, 'dtype': self.dtype, 'experts': self.experts, 'videos': self.videos}, f)

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()


class FeatureStore:
  """Read side: everything is np.memmap, a sample's rows are views (no parsing, no per-video file)."""

  def __init__(self, path):
    with open(os.path.join(path, 'meta.json')) as f:
      meta = json.load(f)
    if meta.get('version') != 1:
      raise ValueError('unknown feature store version %r' % meta.get('version'))
    self.path, self.dtype, self.experts, self.videos = path, meta['dtype'], meta['experts'], meta['videos']
    self.index = {v: i for i, v in enumerate(self.videos)}
    self._rows, self._t, self._off = {}, {}, {}
    for e, dim in self.experts.items():
      off = np.fromfile(os.path.join(path, e + '.off'), dtype=np.int64)
      if off.shape[0] != len(self.videos) + 1:
        raise ValueError('%s.off does not match the video list' % e)
      total = int(off[-1])
      self._off[e] = off
      if total:
        self._rows[e] = np.memmap(os.path.join(path, e + '.rows'), mode='r',
                                  dtype=np.uint16 if self.dtype == 'bf16' else np.float32, shape=(total, dim))
        self._t[e] = np.memmap(os.path.join(path, e + '.t'), mode='r', dtype=np.float64, shape=(total,))
      else:
        self._rows[e] = np.zeros((0, dim), np.uint16 if self.dtype == 'bf16' else np.float32)
        self._t[e] = np.zeros((0,), np.float64)

    self._cap_off = np.fromfile(os.path.join(path, 'captions.off'), dtype=np.int64)
    self._vid_cap = np.fromfile(os.path.join(path, 'captions.vid'), dtype=np.int64)
    ntok = int(self._cap_off[-1])
    self._tok = (np.memmap(os.path.join(path, 'captions.tok'), mode='r', dtype=np.int32, shape=(ntok,)) if ntok
                 else np.zeros((0,), np.int32))

    self._wblob, self._wtimes = b'', np.zeros((0, 2))
    self._wbyte = self._wcount = np.zeros((1,), np.int64)
    self._vid_wcap = np.zeros((len(self.videos) + 1,), np.int64)
    if os.path.exists(os.path.join(path, 'captions.woff')):  # (stores written before r05 have no word-level captions)
      woff = np.fromfile(os.path.join(path, 'captions.woff'), dtype=np.int64).reshape(2, -1)
      self._wbyte, self._wcount = woff[0], woff[1]
      self._vid_wcap = np.fromfile(os.path.join(path, 'captions.wvid'), dtype=np.int64)
      nw = int(self._wcount[-1])
      if nw:
        self._wblob = np.memmap(os.path.join(path, 'captions.words'), mode='r', dtype=np.uint8, shape=(int(self._wbyte[-1]),))
        self._wtimes = np.memmap(os.path.join(path, 'captions.wt'), mode='r', dtype=np.float64, shape=(nw, 2))

  def __len__(self):
    return len(self.videos)

  def has_caption_words(self):
    """Does the store hold word-level captions (`FeatureStoreWriter.add(caption_words=...)`, r05) for any video?"""
    return int(self._vid_wcap[-1]) > 0

  def caption_words(self, i):
    """-> (list of word lists, list of [n_words, 2] second arrays), one entry per stored caption of video i"""
    i = self.index[i] if isinstance(i, str) else i
    a, b = int(self._vid_wcap[i]), int(self._vid_wcap[i + 1])
    words, times = [], []
    for c in range(a, b):
      blob = bytes(self._wblob[int(self._wbyte[c]):int(self._wbyte[c + 1])]).decode('UTF-8')
      n = int(self._wcount[c + 1] - self._wcount[c])
      words.append(blob.split('\x1f') if n else [])
      times.append(np.array(self._wtimes[int(self._wcount[c]):int(self._wcount[c + 1])]))
    return words, times

  def captions(self, i):
    """-> list of int32 token-id views, one per stored caption of video i"""
    i = self.index[i] if isinstance(i, str) else i
    a, b = int(self._vid_cap[i]), int(self._vid_cap[i + 1])
    return [self._tok[self._cap_off[c]:self._cap_off[c + 1]] for c in range(a, b)]

  def rows(self, expert, i):
    """-> ([n, D] rows as stored, [n] mean second of each row) of video i (index or id)"""
    i = self.index[i] if isinstance(i, str) else i
    a, b = self._off[expert][i], self._off[expert][i + 1]
    return self._rows[expert][a:b], self._t[expert][a:b]


# ---- the wire format -------------------------------------------------------------------------------------------------
class RaggedLayout:
  """Byte offsets of one minibatch in the wire buffer: [ind | t] fp32 [M, B, T] each, then X_e for every expert with the
  capacity of a full batch (rows padded to 128, K padded to 128: the GEMM tiles of csrc/gemm2.hip)."""

  def __init__(self, experts, batch, tokens):
    """experts: ordered (name, dim) pairs in the model's modality order"""
    self.experts = [(str(n), int(d)) for n, d in (experts.items() if isinstance(experts, dict) else experts)]
    self.batch, self.tokens = int(batch), int(tokens)
    m = len(self.experts)
    self.rows_pad = _round_up(self.batch * (self.tokens + 1), 128)
    self.ind_off = 0
    self.t_off = m * batch * tokens * 4
    off = _round_up(2 * self.t_off, 256)
    self.header_bytes = off
    self.x_off, self.dpad = {}, {}
    for name, dim in self.experts:
      self.dpad[name] = _round_up(dim, 128)
      self.x_off[name] = off
      off += self.rows_pad * self.dpad[name] * 2
    self.nbytes = off

  def same_as(self, other):
    return self.experts == other.experts and (self.batch, self.tokens) == (other.batch, other.tokens)


class RaggedFeatures:
  """One minibatch of video features in the wire format (host, optionally pinned, or device)."""

  def __init__(self, layout, device='cpu', pin_memory=False):
    self.layout = layout
    dev = torch.device(device)
    self.flat = torch.zeros(layout.nbytes, dtype=torch.uint8, device=dev,
                            pin_memory=bool(pin_memory) and dev.type == 'cpu')
    m, b, t = len(layout.experts), layout.batch, layout.tokens
    n = m * b * t * 4
    ind = self.flat[layout.ind_off:layout.ind_off + n].view(torch.float32).view(m, b, t)
    tt = self.flat[layout.t_off:layout.t_off + n].view(torch.float32).view(m, b, t)
    tt.fill_(1.0)  # choose_or_pad_to_len: padded slots carry t = 1 (base_dataset.py:93)
    self.x, self.ind, self.t = {}, {}, {}
    for i, (name, _) in enumerate(layout.experts):
      o, dp = layout.x_off[name], layout.dpad[name]
      self.x[name] = self.flat[o:o + layout.rows_pad * dp * 2].view(torch.bfloat16).view(layout.rows_pad, dp)
      self.ind[name], self.t[name] = ind[i], tt[i]
    # live rows of every X_e (B pooled rows + valid feature rows): host-side knowledge, the device recounts from `ind`
    self.live = {name: b for name, _ in layout.experts}

  # the model only needs these
  @property
  def device(self):
    return self.flat.device

  @property
  def batch(self):
    return self.layout.batch

  @property
  def tokens(self):
    return self.layout.tokens

  def copy_from(self, src, non_blocking=True):
    """Upload / copy `src` (same layout) into this buffer: the [ind | t] header and the LIVE prefix of every X_e."""
    if not self.layout.same_as(src.layout):
      raise ValueError('RaggedFeatures layouts differ')
    L = self.layout
    self.flat[:L.header_bytes].copy_(src.flat[:L.header_bytes], non_blocking=non_blocking)
    for name, _ in L.experts:
      o, n = L.x_off[name], src.live[name] * L.dpad[name] * 2
      self.flat[o:o + n].copy_(src.flat[o:o + n], non_blocking=non_blocking)
    self.live = dict(src.live)
    return self

  def live_bytes(self):
    """bytes `copy_from` moves for this minibatch"""
    L = self.layout
    return L.header_bytes + sum(self.live[n] * L.dpad[n] * 2 for n, _ in L.experts)

  # ---- conversions (host) ----------------------------------------------------------------------
  @classmethod
  def from_dense(cls, features, features_t, features_ind, features_maxpool, experts=None, pin_memory=False):
    """The reference's collated minibatch (mix_dataset.py:112-144: dicts of (B, T, D) / (B, T) / (B, D) fp32 arrays or
    CPU tensors) -> wire format."""
    names = list(experts) if experts is not None else list(features.keys())
    as_np = lambda v: v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    f0 = as_np(features[names[0]])
    layout = RaggedLayout([(n, as_np(features[n]).shape[-1]) for n in names], f0.shape[0], f0.shape[1])
    out = cls(layout, 'cpu', pin_memory)
    b = layout.batch
    for n in names:
      f, ind = as_np(features[n]).astype(np.float32), as_np(features_ind[n]).astype(np.float32)
      d = f.shape[-1]
      xv = out.x[n].view(torch.int16).numpy().view(np.uint16)
      xv[:b, :d] = to_bf16(as_np(features_maxpool[n]))
      valid = ind.reshape(-1) != 0
      cnt = int(valid.sum())
      xv[b:b + cnt, :d] = to_bf16(f.reshape(-1, d)[valid])
      out.ind[n].copy_(torch.from_numpy(ind))
      out.t[n].copy_(torch.from_numpy(as_np(features_t[n]).astype(np.float32)))
      out.live[n] = b + cnt
    return out

  def to_dense(self):
    """-> (features, features_t, features_ind, features_maxpool) dicts of fp32 CPU tensors in the reference's dense
    layout (feature values as the device sees them: bf16-rounded)."""
    L = self.layout
    b, t = L.batch, L.tokens
    feats, ft, fi, fm = {}, {}, {}, {}
    for n, d in L.experts:
      x = self.x[n].detach().float().cpu()
      ind = self.ind[n].detach().float().cpu()
      valid = (ind.reshape(-1) != 0)
      dense = torch.zeros(b * t, d)
      dense[valid] = x[b:b + int(valid.sum()), :d]
      feats[n], ft[n], fi[n], fm[n] = dense.view(b, t, d), self.t[n].detach().float().cpu().clone(), ind.clone(), x[:b, :d].clone()
    return feats, ft, fi, fm


class RaggedCollator:
  """FeatureStore -> RaggedFeatures for a list of video indices: the video half of `BaseDataset.__getitem__`
  (base/base_dataset.py:765-850) and `MixDataset.collate_data`, writing straight into the wire buffer.

  clip window: `window(i) -> (feat_start, feat_end)` or None for the whole video (clip_duration = inf, every published
  config), in which case feat_start = 0 (base_dataset.py:760-762)."""

  def __init__(self, store, experts, batch, max_expert_tokens, training, temporal_encoding_window=1.0,
               shuffle_feats_t=False, rng=None, pin_memory=False):
    if shuffle_feats_t and training:
      # base_dataset.py:109-110 assigns the RETURN VALUE of rng.shuffle (None) -> NaN timestamps; not reproduced
      raise NotImplementedError('shuffle_feats_t: the reference writes NaN timestamps in this mode')
    self.store, self.training, self.rng = store, bool(training), rng
    self.window_len = float(temporal_encoding_window)
    names = list(experts)
    for n in names:
      if n not in store.experts:
        raise KeyError('expert %r is not in the store' % n)
    self.layout = RaggedLayout([(n, store.experts[n]) for n in names], batch, max_expert_tokens)
    self.pin_memory = pin_memory

  def new_buffer(self):
    return RaggedFeatures(self.layout, 'cpu', self.pin_memory)

  def collate_tokens(self, indices, captions_per_video, max_text_words, pad_caption, sep_id=None):
    """token_ids [B, C, W, 2] int32 and query_masks [B, C] int32 as `__getitem__` + `collate_data` build them for
    query_shuffling='indiv', caption_length=inf, n_pairs=1 (base_dataset.py:594-600, 647-668, 741-757, 852-858;
    mix_dataset.py:134-136): the first C stored captions of every video, missing ones replaced by `pad_caption` (the
    token ids of the reference's filler caption "0", :661-664) with query mask 0."""
    tok = np.zeros((len(indices), captions_per_video, max_text_words, 2))
    qm = np.zeros((len(indices), captions_per_video))
    for s, i in enumerate(indices):
      caps = self.store.captions(i)
      for c in range(captions_per_video):
        have = c < len(caps)
        tok[s, c] = crop_or_pad_tokens(caps[c] if have else pad_caption, max_text_words, sep_id)
        qm[s, c] = 1.0 if have else 0.0
    return torch.from_numpy(tok.astype(np.int32)), torch.from_numpy(qm.astype(np.int32))

  def collate_captions(self, indices, captions_per_video, max_text_words, tokenizer, query_shuffling='indiv',
                       caption_length=float('inf'), clip_duration=float('inf'), restrict_test_captions=None,
                       py_random=None, remove_stop_words=False, n_pairs=1):
    """The caption half of `BaseDataset.__getitem__` (base/base_dataset.py:569-757) on the store's word-level captions, for
    the caption sampling modes of the reference's published configurations (`remove_stop_words=False`, `n_pairs=1`:
    utils/util.py:431, base/base_dataset.py:165,732-734 -- either option set raises NotImplementedError instead of silently
    producing other tokens and feature windows than the reference would):
      query_shuffling  'indiv' (each caption on its own), 'cat' (all captions concatenated in order), 'shuf' (shuffled,
                       then concatenated), 'shufk<N>' (shuffled, the first N concatenated)               :594-625
      caption_length   inf, n or [min, max]: a window of that many consecutive "sentences" (= words: every word is its
                       own sentence, :653-660) at a random start                                          :686-724
      clip_duration    inf, seconds or [min, max]: returns the feature window centred on the kept words     :701-712, 757-767
      words after 500 s dropped (:657), an empty caption becomes ".", a missing one the filler "0" with query mask 0
      (:660-668), cut to max_text_words, tokenised by `tokenizer` as `tokenize_caption` does (:320-346: joined, a period
      appended, capitalised, [CLS] ... [SEP], cut, last token forced to [SEP]).
    Randomness is consumed exactly as the reference consumes it: the caption shuffles from Python's `random` (py_random,
    default the global module), window lengths / clip lengths / window starts from `np.random` in training and from
    RandomState(idx) per sample otherwise.  tokenizer: .tokenize(str) -> tokens, .convert_tokens_to_ids(tokens) -> ids,
    .cls_token, .sep_token (a HuggingFace tokenizer).
    -> token_ids [B, C, W, 2] int32, query_masks [B, C] int32, windows [(feat_start, feat_end)] per sample."""
    import random as _random
    import re
    if remove_stop_words:
      raise NotImplementedError('collate_captions: remove_stop_words=True (base/base_dataset.py:118-147, 732-734) is not '
                                'ported -- the published configurations leave it off (utils/util.py:431)')
    if n_pairs != 1:
      raise NotImplementedError('collate_captions: n_pairs = %r (base/base_dataset.py:569-757 with several clip / caption '
                                'pairs per item); the published configurations use 1' % (n_pairs,))
    if not self.store.has_caption_words():
      raise ValueError('collate_captions: this feature store holds no word-level captions (written before r05, or without '
                       'captions=): rebuild it with FeatureStoreWriter(..., captions=...)')
    pyr = py_random if py_random is not None else _random
    C, W = captions_per_video, max_text_words
    z = re.match(r'shufk(\d*)', query_shuffling)
    if query_shuffling not in ('indiv', 'cat', 'shuf') and not z:
      raise ValueError('query_shuffling: indiv | cat | shuf | shufk<N>')
    tok = np.zeros((len(indices), C, W, 2))
    qm = np.zeros((len(indices), C))
    windows = []
    for s, i in enumerate(indices):
      idx = self.store.index[i] if isinstance(i, str) else int(i)
      captions, captions_t = self.store.caption_words(idx)
      if not captions:
        raise ValueError('video %r has no word-level captions in the store' % (i,))
      captions = [np.array(c, dtype=object) for c in captions]
      if restrict_test_captions and self.store.videos[idx] in restrict_test_captions:  # :577-582
        keep = restrict_test_captions[self.store.videos[idx]]
        captions, captions_t = [captions[keep]], [captions_t[keep]]
      raw, raw_t = [], []
      for cap_nb in range(min(len(captions), C)):  # :592-625
        if query_shuffling == 'indiv':
          raw.append(captions[cap_nb]); raw_t.append(captions_t[cap_nb])
        elif query_shuffling == 'cat':
          raw.append(np.concatenate(captions)); raw_t.append(np.concatenate(captions_t))
        else:
          c = list(zip(captions, captions_t))
          pyr.shuffle(c)
          captions, captions_t = zip(*c)
          nb_keep = min(int(z.groups()[0]), len(captions)) if z else len(captions)
          raw.append(np.concatenate(captions[:nb_keep])); raw_t.append(np.concatenate(captions_t[:nb_keep]))
      sentences_of = []
      for cap_idx in range(C):  # :648-668
        if cap_idx < len(raw):
          cap, cap_t = np.array([str(w) for w in raw[cap_idx]]), np.array(raw_t[cap_idx]).reshape(-1, 2)
          keep_ids = cap_t[:, 0] < 500
          cap, cap_t = np.expand_dims(cap[keep_ids], axis=-1), np.expand_dims(cap_t[keep_ids], axis=-1)
          if len(cap) < 1:
            cap, cap_t = np.array([['.']]), np.array([[[0, 0]]])
        else:
          cap, cap_t = np.array([['0']]), np.array([[[0, 0]]])
        sentences_of.append((cap, cap_t))
      qm[s, :len(raw)] = 1
      clip_length, selected_t = float('inf'), None
      for cap_idx in range(C):  # :679-748
        rng = np.random if self.training else np.random.RandomState(idx)
        lo, hi = (caption_length if isinstance(caption_length, (list, tuple)) else (caption_length, caption_length))
        nb_sentences = float('inf') if lo == float('inf') else rng.randint(lo, hi + 1)
        clo, chi = (clip_duration if isinstance(clip_duration, (list, tuple)) else (clip_duration, clip_duration))
        clip_length = float('inf') if chi == float('inf') else rng.uniform(clo, chi)
        sentences, sentences_t = sentences_of[cap_idx]
        nb_sentences = min(nb_sentences, len(sentences))
        choice = rng.randint(len(sentences) + 1 - nb_sentences)
        selected = np.concatenate(sentences[choice:choice + nb_sentences])[:W]
        selected_t = np.concatenate(sentences_t[choice:choice + nb_sentences])[:W]
        tokens = [tokenizer.cls_token] + list(tokenizer.tokenize(caption_text(list(selected)))) + [tokenizer.sep_token]
        tokens = tokens[:W]
        tokens[-1] = tokenizer.sep_token
        ids = tokenizer.convert_tokens_to_ids(tokens)
        tok[s, cap_idx, :len(ids), 0] = ids
        tok[s, cap_idx, :len(ids), 1] = 1
      if clip_length == float('inf'):  # :757-767 (the window follows the LAST caption of the sample, as in the reference)
        windows.append((0.0, float('inf')))
      else:
        c_time = np.mean((np.min(selected_t), np.max(selected_t)))
        windows.append((c_time - clip_length / 2, c_time - clip_length / 2 + clip_length))
    return torch.from_numpy(tok.astype(np.int32)), torch.from_numpy(qm.astype(np.int32)), windows

  def collate(self, indices, out=None, window=None):
    """Rows are drawn SAMPLE-major -- for every sample, every expert in turn -- as `BaseDataset.__getitem__` does
    (base/base_dataset.py:772-833), so a seeded `rng` is consumed sample by sample like the reference's np.random.
    (The reference iterates `self.experts`, a Python set: its expert order inside a sample depends on the process' hash
    seed, so an identical draw SEQUENCE over a whole batch cannot be pinned; each draw is `choose_or_pad_to_len`'s.)
    Inside DataLoader workers pass a per-worker RandomState: the global np.random is duplicated across forked workers."""
    L = self.layout
    if len(indices) != L.batch:
      raise ValueError('expected %d samples, got %d' % (L.batch, len(indices)))
    out = self.new_buffer() if out is None else out
    b, T = L.batch, L.tokens
    f32 = self.store.dtype == 'f32'
    views, cursor = {}, {}
    for n, d in L.experts:
      xv = out.x[n].view(torch.int16).numpy().view(np.uint16)
      ind, tt = out.ind[n].numpy(), out.t[n].numpy()
      ind[:] = 0.0
      tt[:] = 1.0
      xv[:b] = 0
      views[n] = (xv, ind, tt)
      cursor[n] = b
    for s, i in enumerate(indices):
      start, end = (0.0, float('inf')) if window is None else window(i)
      for n, d in L.experts:
        xv, ind, tt = views[n]
        rows, sec = self.store.rows(n, i)
        if rows.shape[0] and window is not None:
          sel = np.nonzero(np.logical_and(start <= sec, sec <= end))[0]  # base_dataset.py:780-782
        else:
          sel = None
        count = rows.shape[0] if sel is None else sel.shape[0]
        if count == 0:
          continue  # missing expert: zero max-pool row, ind = 0, t = 1 (base_dataset.py:797-803)
        pick = choose_rows(count, T, self.training, self.rng)
        src = pick if sel is None else sel[pick]
        keep = src.shape[0]
        window_rows = rows if sel is None else rows[sel]
        cur = cursor[n]
        if f32:
          xv[s, :d] = to_bf16(np.max(window_rows, axis=0))
          xv[cur:cur + keep, :d] = to_bf16(rows[src])
        else:  # max over bf16 values == bf16(max over the fp32 values): the rounding is monotone
          xv[s, :d] = to_bf16(np.max(from_bf16(window_rows), axis=0))
          xv[cur:cur + keep, :d] = rows[src]
        ind[s, :keep] = 1.0
        tt[s, :keep] = ((sec[src] - start) / self.window_len + 2).astype(np.float32)  # base_dataset.py:773-776
        cursor[n] = cur + keep
    for n, _ in L.experts:
      out.live[n] = cursor[n]
    return out
