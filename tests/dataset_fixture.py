"""TEST INFRASTRUCTURE: synthetic per-video feature files for the input-pipeline tests (what the reference keeps in one h5
file per video, base/base_dataset.py:426-497), generated from a seed so that the golden fixture
tests/golden/dataset_items.npz only has to hold the reference's OUTPUT."""
import numpy as np

DIMS = {'s3d': 64, 'vggish': 32, 'rgb': 48, 'face': 16}   # raw_input_dims handed to the reference dataset (order = experts)
MAX_TOKENS, MAX_WORDS, WINDOW = 8, 10, 1.0
CLIP_SECONDS = 6.0


def make_videos(seed=5):
  """-> list of (vid, h5_like dict): keys as in the reference's files, values numpy arrays"""
  rng = np.random.RandomState(seed)
  rows = {'s3d': [12, 5, 8, 0, 20, 3, 9, 1], 'vggish': [12, 5, 0, 7, 20, 3, 8, 2], 'rgb': [30, 4, 8, 9, 0, 6, 7, 5],
          'face': [3, 0, 8, 2, 1, 0, 10, 4]}
  videos = []
  for v in range(8):
    h5 = {}
    for e, d in DIMS.items():
      n = rows[e][v]
      if v == 3 and e == 's3d':
        continue                                      # no key at all
      x = rng.randn(n, d).astype(np.float32)
      if v == 5 and e == 'rgb':
        x[0, 0] = np.nan                              # flagged invalid by its first value (:470)
      h5['features.' + e] = x
      if e in ('s3d', 'vggish') and n:
        start = np.cumsum(rng.uniform(0.5, 1.5, size=n + (2 if v == 1 else 0)))  # v1: more timings than rows (:480)
        h5['features_t.' + e] = np.stack([start, start + 1.0], axis=-1)
      if e == 'rgb' and v == 2:
        h5['features_t.' + e] = np.zeros((n, 2))      # ignored: only s3d / vggish timings are read (:476)
    words = ['w%d' % k for k in range(3 + v)]
    h5['raw_captions.0'] = np.array(words, dtype=object)
    h5['raw_captions_t.0'] = np.stack([np.arange(len(words)) * 1.5 + 1.0, np.arange(len(words)) * 1.5 + 2.0], axis=-1)
    videos.append(('video%d' % v, h5))
  return videos


def clip_window(h5, clip_seconds=CLIP_SECONDS, max_words=MAX_WORDS):
  """feat_start / feat_end of base_dataset.py:752-763 for one caption, clip_duration = clip_seconds, caption_length = inf."""
  # the reference concatenates the per-word [start, end] pairs into one flat list BEFORE cutting it to max_text_words
  # values (base_dataset.py:731, :738): the window is centred on the first max_words / 2 words
  t = h5['raw_captions_t.0'].reshape(-1)[:max_words]
  c = np.mean((np.min(t), np.max(t)))
  return c - clip_seconds / 2, c - clip_seconds / 2 + clip_seconds


def h5_features(h5):
  feats = {k[len('features.'):]: v for k, v in h5.items() if k.startswith('features.')}
  times = {k[len('features_t.'):]: v for k, v in h5.items() if k.startswith('features_t.')}
  return feats, times


def make_caption_videos(seed=9):
  """Videos with SEVERAL captions each (raw_captions.<i> / raw_captions_t.<i>, base_dataset.py:446-458) for the caption
  sampling modes: captions of 2..6 words, word times spread over the video; video 2 has a caption spoken entirely after
  500 s (dropped: the caption becomes "."), video 4 a single caption (a second one requested -> the filler "0"), video 5
  three captions of EQUAL length (what 'shufk' needs under a NumPy that refuses ragged arrays: the reference stacks the
  selected captions' times with np.array, :627)."""
  rng = np.random.RandomState(seed)
  videos = make_videos()
  out = []
  for v, (vid, h5) in enumerate(videos):
    h5 = {k: val for k, val in h5.items() if not k.startswith('raw_captions')}
    ncap = [3, 2, 2, 3, 1, 3, 2, 4][v]
    for c in range(ncap):
      n = 3 if v == 5 else 2 + (v + 2 * c) % 5
      words = ['V%dc%dw%d' % (v, c, k) if (k + c) % 3 else 'x%d.' % k for k in range(n)]
      start = np.cumsum(rng.uniform(0.4, 2.0, size=n)) + 3.0 * c
      if v == 2 and c == 1:
        start = start + 600.0
      h5['raw_captions.%d' % c] = np.array(words, dtype=object)
      h5['raw_captions_t.%d' % c] = np.stack([start, start + 0.5], axis=-1)
    out.append((vid, h5))
  return out


def h5_captions(h5):
  n = len([k for k in h5 if k.startswith('raw_captions.')])
  return [list(h5['raw_captions.%d' % c]) for c in range(n)], [h5['raw_captions_t.%d' % c] for c in range(n)]
