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
