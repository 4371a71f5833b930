"""Input pipeline: memory-mapped store + ragged wire format against what the REAL reference
pipeline produced on the same synthetic videos (tests/golden/dataset_items.npz, oracle/gen_dataset_golden.py:
base/base_dataset.py get_sample_data / __getitem__ + mix_dataset.py collate_data)."""
import numpy as np
import pytest
import torch

from mmt_amd import feature_store as FS
from tests import dataset_fixture as DF
from tests.fixtures import load_npz


class FixtureTokenizer:
  """the stand-in tokenizer oracle/gen_dataset_golden.py gave the reference dataset (any tokenizer would do: the store
  keeps what it returns)"""
  CLS, SEP = 101, 102

  def ids(self, words):
    toks = FS.caption_text(words).replace('.', ' .').split()
    return [self.CLS] + [1000 + sum(map(ord, t)) % 997 for t in toks] + [self.SEP]


def build_store(path, dtype):
  tok = FixtureTokenizer()
  with FS.FeatureStoreWriter(str(path), DF.DIMS, dtype=dtype) as w:
    for vid, h5 in DF.make_videos():
      feats, times = DF.h5_features(h5)
      w.add(vid, feats, times, captions=[tok.ids(list(h5['raw_captions.0']))])
  return FS.FeatureStore(str(path))


def bf16_round(a):
  return FS.from_bf16(FS.to_bf16(a)).reshape(np.shape(a))


def test_to_bf16_is_torch_round_to_nearest_even():
  g = torch.Generator().manual_seed(0)
  x = torch.randn(100_000, generator=g) * torch.logspace(-20, 20, 100_000)
  x[:6] = torch.tensor([0.0, -0.0, 1.0, 1.00390625, 1.01171875, 3.3895e38])  # ties to even both ways, near-overflow
  want = x.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
  assert np.array_equal(FS.to_bf16(x.numpy()), want)
  assert np.array_equal(FS.from_bf16(want), x.to(torch.bfloat16).float().numpy())


def test_feature_timings_match_reference():
  g = load_npz('dataset_items')
  assert np.array_equal(FS.feature_timings(7, FS.FEAT_WIDTH.get('rgb')), g['timings/rgb7'])
  assert np.array_equal(FS.feature_timings(4, FS.FEAT_WIDTH.get('face')), g['timings/face4'])
  assert np.array_equal(FS.feature_timings(6, 1.0, stride=2.0, group=2), g['timings/group'])


def test_training_row_choice_matches_reference_function():
  """choose_or_pad_to_len under np.random.seed(s) (base_dataset.py:96-113) == choose_rows with RandomState(s)."""
  for row in load_npz('dataset_items')['train_choice']:
    n, seed = int(row[0]), int(row[1])
    T = DF.MAX_TOKENS
    feat, t, ind = row[2:2 + T], row[2 + T:2 + 2 * T], row[2 + 2 * T:]
    pick = FS.choose_rows(n, T, True, np.random.RandomState(seed))
    keep = min(n, T)
    assert np.array_equal(pick, feat[:keep].astype(np.int64))       # the fixture's feature value IS the row index
    assert np.array_equal(pick * 0.5, t[:keep]) and np.all(t[keep:] == 1) and ind.sum() == keep
    np.random.seed(seed)
    assert np.array_equal(FS.choose_rows(n, T, True), pick)           # default generator = the reference's global one


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
@pytest.mark.parametrize('tag', ['full', 'clip'])
def test_collated_minibatch_equals_reference_pipeline(tmp_path, dtype, tag):
  g = load_npz('dataset_items')
  store = build_store(tmp_path / 's', dtype)
  videos = dict(DF.make_videos())
  coll = FS.RaggedCollator(store, list(DF.DIMS), len(store), DF.MAX_TOKENS, training=False,
                           temporal_encoding_window=DF.WINDOW)
  window = None if tag == 'full' else (lambda i: DF.clip_window(videos[store.videos[i]]))
  rag = coll.collate(list(range(len(store))), window=window)
  feats, ft, fi, fm = rag.to_dense()
  for e in DF.DIMS:
    assert np.array_equal(fi[e].numpy(), g['%s/features_ind/%s' % (tag, e)]), e
    assert np.array_equal(ft[e].numpy(), g['%s/features_t/%s' % (tag, e)]), e            # fp32, bit for bit
    assert np.array_equal(feats[e].numpy(), bf16_round(g['%s/features/%s' % (tag, e)])), e
    assert np.array_equal(fm[e].numpy(), bf16_round(g['%s/features_maxpool/%s' % (tag, e)])), e
    assert rag.live[e] == len(store) + int(g['%s/features_ind/%s' % (tag, e)].sum())
  # the same minibatch through the dense door (a reference loader's output handed over as is)
  dense = {k: {e: g['%s/%s/%s' % (tag, k, e)] for e in DF.DIMS} for k in
           ('features', 'features_t', 'features_ind', 'features_maxpool')}
  rag2 = FS.RaggedFeatures.from_dense(dense['features'], dense['features_t'], dense['features_ind'],
                                      dense['features_maxpool'], experts=list(DF.DIMS))
  assert rag2.live == rag.live
  L = rag.layout
  assert torch.equal(rag2.flat[:L.header_bytes], rag.flat[:L.header_bytes])
  for e in DF.DIMS:
    assert torch.equal(rag2.x[e][:rag.live[e]].view(torch.int16), rag.x[e][:rag.live[e]].view(torch.int16)), e


def test_store_edge_cases_and_views(tmp_path):
  store = build_store(tmp_path / 's', 'bf16')
  assert store.videos == ['video%d' % i for i in range(8)] and len(store) == 8
  assert store.rows('s3d', 3)[0].shape == (0, 64)          # key absent
  assert store.rows('rgb', 5)[0].shape == (0, 48)          # NaN first value
  assert store.rows('rgb', 4)[0].shape == (0, 48)          # empty array
  assert store.rows('vggish', 'video1')[0].shape == (5, 32) and store.rows('vggish', 1)[1].shape == (5,)
  assert store.rows('rgb', 2)[1][1] == pytest.approx(0.3)  # stored rgb timings are ignored: 0.2 s per row, mean of [0.2, 0.4]
  assert np.all(store.rows('face', 2)[1] == -1.0)
  rows, _ = store.rows('s3d', 0)
  assert isinstance(rows, np.memmap) or isinstance(rows.base, np.memmap)   # a view, nothing parsed or copied
  with pytest.raises(ValueError):
    FS.FeatureStoreWriter(str(tmp_path / 'bad'), {'s3d': 64}).add('v', {'s3d': np.zeros((3, 5), np.float32)})
  with pytest.raises(KeyError):
    FS.RaggedCollator(store, ['s3d', 'scene'], 2, 8, False)
  with pytest.raises(NotImplementedError):
    FS.RaggedCollator(store, ['s3d'], 2, 8, True, shuffle_feats_t=True)


def test_collator_reuses_a_buffer_and_training_is_a_valid_draw(tmp_path):
  store = build_store(tmp_path / 's', 'bf16')
  coll = FS.RaggedCollator(store, ['s3d', 'rgb'], 4, DF.MAX_TOKENS, training=True, rng=np.random.RandomState(1))
  buf = coll.new_buffer()
  a = coll.collate([0, 4, 0, 4], out=buf)
  assert a is buf
  feats, ft, fi, fm = a.to_dense()
  full, sec = store.rows('s3d', 0)
  full = FS.from_bf16(full)
  for s in (0, 2):   # 12 rows -> 8 kept, in increasing time, each an actual row of the video
    idx = [int(np.nonzero((full == feats['s3d'][s, k].numpy()).all(1))[0][0]) for k in range(8)]
    assert idx == sorted(idx) and len(set(idx)) == 8
    assert np.allclose(ft['s3d'][s].numpy(), (sec[idx] / 1.0 + 2).astype(np.float32))
    assert np.array_equal(fm['s3d'][s].numpy(), full.max(0))     # pooled over ALL rows, not just the kept ones
  assert not torch.equal(feats['s3d'][0], feats['s3d'][2])        # two independent draws
  assert fi['rgb'][1].sum() == 0 and fm['rgb'][1].abs().sum() == 0 and torch.all(ft['rgb'][1] == 1)
  b = coll.collate([1, 1, 1, 1], out=buf)                         # a smaller batch leaves nothing of the old one visible
  feats2, _, fi2, _ = b.to_dense()
  assert fi2['s3d'].sum() == 20 and b.live['s3d'] == 4 + 20 and b.live_bytes() < a.layout.nbytes


def test_collated_token_ids_equal_reference_pipeline(tmp_path):
  """token_ids (B, C, W, 2) as the real `__getitem__` + `collate_data` built them: captions of 3..10 words against
  max_text_words = 10 (the longest is cut and its last kept token forced to [SEP])."""
  g = load_npz('dataset_items')
  store = build_store(tmp_path / 's', 'bf16')
  coll = FS.RaggedCollator(store, list(DF.DIMS), len(store), DF.MAX_TOKENS, training=False)
  tok = FixtureTokenizer()
  ids, qm = coll.collate_tokens(list(range(len(store))), 1, DF.MAX_WORDS, pad_caption=tok.ids(['0']), sep_id=tok.SEP)
  assert ids.dtype == torch.int32 and np.array_equal(ids.numpy(), g['full/token_ids'])
  assert np.array_equal(ids.numpy(), g['clip/token_ids']) and int(qm.sum()) == len(store)
  assert int(ids[7, 0, :, 1].sum()) == DF.MAX_WORDS and int(ids[7, 0, -1, 0]) == tok.SEP   # 10 words: cut, [SEP] forced
  assert int(ids[0, 0, :, 1].sum()) == 6                                                    # [CLS] w0 w1 w2 . [SEP]
  # more captions requested than stored: the filler caption with query mask 0 (base_dataset.py:661-664)
  ids2, qm2 = coll.collate_tokens([0, 1], 2, DF.MAX_WORDS, pad_caption=tok.ids(['0']), sep_id=tok.SEP)
  assert qm2.tolist() == [[1, 0], [1, 0]] and ids2[0, 1, :, 1].sum() == 4 and torch.equal(ids2[:, 0], ids[:2, 0])
  assert [len(c) for c in store.captions('video3')] == [3 + 3 + 3]


class FixtureHFTokenizer:
  """the stand-in oracle/gen_dataset_golden.py gave the reference dataset, with a HuggingFace tokenizer's surface"""
  cls_token, sep_token = '[CLS]', '[SEP]'

  def tokenize(self, text):
    return text.replace('.', ' .').split()

  def convert_tokens_to_ids(self, tokens):
    return [101 if t == '[CLS]' else 102 if t == '[SEP]' else 1000 + sum(map(ord, t)) % 997 for t in tokens]


def build_caption_store(path):
  with FS.FeatureStoreWriter(str(path), DF.DIMS, dtype='f32') as w:
    for vid, h5 in DF.make_caption_videos():
      feats, times = DF.h5_features(h5)
      words, wt = DF.h5_captions(h5)
      w.add(vid, feats, times, caption_words=words, caption_times=wt)
  return FS.FeatureStore(str(path))


@pytest.mark.parametrize('case', ['cat2', 'shuf2', 'shufk2', 'indiv_window', 'cat_window_clip', 'train_window', 'train_shuf_clip'])
def test_caption_sampling_modes_equal_reference_pipeline(tmp_path, case):
  """`RaggedCollator.collate_captions` against the REAL `BaseDataset.__getitem__` + `collate_data` (oracle/gen_dataset_golden.py
  -> tests/golden/dataset_captions.npz) for every caption sampling mode of the reference: captions concatenated in order /
  shuffled / shuffled and cut to the first N (base_dataset.py:594-625), windows of consecutive words of random length and
  start (:686-724), words after 500 s dropped, empty and missing captions (:657-668), in evaluation (RandomState(idx)) and in
  training (the global generators, seeded per item as the generator script seeded them); token ids and query masks equal
  bit for bit, and where a clip duration is set the feature window the kept words centre selects the reference's rows."""
  import random
  from oracle.gen_dataset_golden import CAPTION_CASES
  g = load_npz('dataset_captions')
  training, cpv, shuffling, cap_len, clip = CAPTION_CASES[case]
  store = build_caption_store(tmp_path / 'c')
  coll = FS.RaggedCollator(store, list(DF.DIMS), len(store), DF.MAX_TOKENS, training=training,
                           temporal_encoding_window=DF.WINDOW)
  toks, masks, windows = [], [], []
  for i in range(len(store)):
    random.seed(1000 + i)
    np.random.seed(2000 + i)
    t, q, w = coll.collate_captions([i], cpv, DF.MAX_WORDS, FixtureHFTokenizer(), query_shuffling=shuffling,
                                    caption_length=cap_len, clip_duration=clip)
    toks.append(t); masks.append(q); windows.append(w[0])
  toks, masks = torch.cat(toks).numpy(), torch.cat(masks).numpy()
  assert toks.dtype == np.int32 and np.array_equal(toks, g[case + '/token_ids'])
  assert np.array_equal(masks, g[case + '/query_masks'])
  if case in ('cat2', 'shuf2'):
    # as many concatenations as min(stored captions, requested): video 4 stores ONE caption (base_dataset.py:592)
    assert masks[4].tolist() == [1, 0] and masks.sum() == 15
  if case == 'cat_window_clip':
    rag = coll.collate(list(range(len(store))), window=lambda i: windows[i])
    _, ft, fi, _ = rag.to_dense()
    for e in DF.DIMS:
      assert np.array_equal(fi[e].numpy(), g['%s/features_ind/%s' % (case, e)]), e
      assert np.array_equal(ft[e].numpy(), g['%s/features_t/%s' % (case, e)]), e
  else:
    assert all(w == (0.0, float('inf')) for w in windows) or clip != float('inf')


def test_caption_words_round_trip_and_edge_cases(tmp_path):
  store = build_caption_store(tmp_path / 'c')
  videos = dict(DF.make_caption_videos())
  for vid in ('video0', 'video4', 'video7'):
    words, times = store.caption_words(vid)
    want_w, want_t = DF.h5_captions(videos[vid])
    assert words == [[str(x) for x in c] for c in want_w]
    assert all(np.array_equal(a, b) for a, b in zip(times, want_t))
  coll = FS.RaggedCollator(store, list(DF.DIMS), len(store), DF.MAX_TOKENS, training=False)
  tok = FixtureHFTokenizer()
  # video 2's second caption lies after 500 s: it becomes "." (base_dataset.py:660-664); video 4 has one caption: the filler "0"
  ids, qm, _ = coll.collate_captions([2, 4], 2, DF.MAX_WORDS, tok)
  assert qm.tolist() == [[1, 1], [1, 0]]
  assert ids[0, 1, :, 0].tolist()[:3] == tok.convert_tokens_to_ids(['[CLS]', '.', '[SEP]']) and int(ids[0, 1, :, 1].sum()) == 3
  assert ids[1, 1, :, 0].tolist()[:4] == tok.convert_tokens_to_ids(['[CLS]', '0', '.', '[SEP]'])
  with pytest.raises(ValueError):
    coll.collate_captions([0], 1, DF.MAX_WORDS, tok, query_shuffling='zigzag')
  old = FS.FeatureStore(str(tmp_path / 'c'))
  assert len(old.captions(0)) == 0  # (this store holds word-level captions only)


def test_collate_captions_rejects_what_it_does_not_port(tmp_path):
  """`remove_stop_words=True` / `n_pairs > 1` (base/base_dataset.py:118-147, 732-734) are not ported -- they
  must raise, not silently yield other tokens than the reference; a store without word-level captions says so up front."""
  with FS.FeatureStoreWriter(str(tmp_path / 's'), {'vggish': 128}) as w:
    w.add('v0', {'vggish': np.zeros((3, 128), np.float32)}, captions=[[101, 5, 102]])
  col = FS.RaggedCollator(FS.FeatureStore(str(tmp_path / 's')), ['vggish'], 1, 4, training=False)
  with pytest.raises(NotImplementedError, match='remove_stop_words'):
    col.collate_captions([0], 1, 8, tokenizer=None, remove_stop_words=True)
  with pytest.raises(NotImplementedError, match='n_pairs'):
    col.collate_captions([0], 1, 8, tokenizer=None, n_pairs=2)
  with pytest.raises(ValueError, match='no word-level captions'):
    col.collate_captions([0], 1, 8, tokenizer=None)
