"""TEST INFRASTRUCTURE -- generates tests/golden/dataset_items.npz by running the REAL reference input pipeline
(base/base_dataset.py `get_sample_data` :426-497 reading per-video h5 files, `__getitem__` :569-896,
data_loader/mix_dataset.py `collate_data` :112-144) on the synthetic videos of tests/dataset_fixture.py.

h5py is not installed: the reference's `h5py.File(path, "r")` is served by an in-memory stand-in exposing exactly what
the reference touches (`.keys()`, `[key].value`, context manager).  Run in the build container only:
    python -m oracle.gen_dataset_golden
"""
import importlib
import os
import sys

import numpy as np

from oracle.ref_loader import load_reference
from tests import dataset_fixture as DF

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'dataset_items.npz')


class _Value:
  def __init__(self, v):
    self.value = v


class _FakeH5:
  files = {}

  def __init__(self, path, mode='r'):
    self._d = self.files[os.path.basename(path)[:-3]]

  def keys(self):
    return self._d.keys()

  def __getitem__(self, k):
    return _Value(self._d[k])

  def __contains__(self, k):
    return k in self._d

  def __enter__(self):
    return self

  def __exit__(self, *a):
    return False


class _Tokenizer:
  cls_token, sep_token = '[CLS]', '[SEP]'

  def tokenize(self, text):
    return text.replace('.', ' .').split()

  def convert_tokens_to_ids(self, tokens):
    return [101 if t == '[CLS]' else 102 if t == '[SEP]' else 1000 + sum(map(ord, t)) % 997 for t in tokens]


def reference_dataset(training, clip_duration, videos=None):
  load_reference()
  sys.modules['h5py'].File = _FakeH5
  BD = importlib.import_module('base.base_dataset')
  cls = type('SyntheticDataset', (BD.BaseDataset,), dict(configure_train_test_splits=lambda s, *a: None,
                                                         sanity_checks=lambda s: None, load_features=lambda s: None))
  ds = cls.__new__(cls)
  videos = DF.make_videos() if videos is None else videos
  _FakeH5.files = dict(videos)
  ds.train, ds.experts, ds.raw_input_dims = training, list(DF.DIMS), dict(DF.DIMS)   # (a list: fixed expert order)
  ds.vid_list = [v for v, _ in videos]
  ds.video_paths = list(ds.vid_list)
  ds.num_train = len(videos)
  ds.reading_from, ds.loaded_in_ram, ds.cache_dir = 'mult_h5', False, '/nonexistent'
  ds.expert_timings = importlib.import_module('utils.expert_timings').expert_timings
  ds.restrict_test_captions, ds.captions_per_video, ds.query_shuffling = None, 1, 'indiv'
  ds.n_pairs, ds.caption_length, ds.clip_duration = 1, float('Inf'), clip_duration
  ds.remove_stop_words, ds.max_text_words, ds.max_expert_tokens = False, DF.MAX_WORDS, DF.MAX_TOKENS
  ds.tokenizer, ds.temporal_encoding_window, ds.shuffle_feats_t = _Tokenizer(), DF.WINDOW, False
  ds.dataset_name = 'Synthetic'
  return ds, BD


def collate(ds, items):
  MD = importlib.import_module('data_loader.mix_dataset')
  holder = type('H', (), {'experts': ds.experts})()
  return MD.MixDataset.collate_data(holder, items)


def main():
  out = {}
  for tag, clip in (('full', float('Inf')), ('clip', DF.CLIP_SECONDS)):
    ds, BD = reference_dataset(False, clip)
    mb = collate(ds, [ds[i] for i in range(len(ds.vid_list))])
    for key in ('features', 'features_t', 'features_ind', 'features_maxpool'):
      for e in DF.DIMS:
        out['%s/%s/%s' % (tag, key, e)] = mb[key][e]
    out[tag + '/token_ids'] = mb['token_ids']
  # training-mode row choice: the reference function itself under a seeded global generator
  ds, BD = reference_dataset(True, float('Inf'))
  picks = []
  for n, seed in ((20, 3), (9, 4), (8, 5), (3, 6)):
    np.random.seed(seed)
    f = np.arange(n, dtype=np.float64)[:, None]
    tensor, tensor_t, ind = BD.choose_or_pad_to_len(f, np.arange(n) * 0.5, DF.MAX_TOKENS, True)
    picks.append(np.concatenate([[n, seed], tensor[:, 0], tensor_t, ind]))
  out['train_choice'] = np.stack(picks)
  out['timings/rgb7'] = ds.get_feature_timings(7, **ds.expert_timings['rgb'])
  out['timings/face4'] = ds.get_feature_timings(4, **ds.expert_timings['face'])
  out['timings/group'] = ds.get_feature_timings(6, 1.0, stride=2.0, group=2)
  np.savez_compressed(OUT, **out)
  print('wrote', OUT, os.path.getsize(OUT), 'bytes')
  captions_golden()


# ---- the caption sampling modes (r05): tests/golden/dataset_captions.npz -------------------------------------------------
OUT_CAP = os.path.join(os.path.dirname(OUT), 'dataset_captions.npz')
# name -> (training, captions_per_video, query_shuffling, caption_length, clip_duration); the seeds a case uses per item are
# random.seed(1000 + idx) (the shuffles) and np.random.seed(2000 + idx) (training-mode draws), set right before __getitem__
CAPTION_CASES = {
    'cat2': (False, 2, 'cat', float('Inf'), float('Inf')),
    'shuf2': (False, 2, 'shuf', float('Inf'), float('Inf')),
    'shufk2': (False, 1, 'shufk2', float('Inf'), float('Inf')),
    'indiv_window': (False, 1, 'indiv', [2, 4], float('Inf')),
    'cat_window_clip': (False, 2, 'cat', 3, [4.0, 8.0]),
    'train_window': (True, 1, 'indiv', [1, 3], float('Inf')),
    'train_shuf_clip': (True, 2, 'shuf', [2, 5], 6.0),
}


def captions_golden():
  import random
  out = {}
  videos = DF.make_caption_videos()
  for name, (training, cpv, shuffling, cap_len, clip) in CAPTION_CASES.items():
    ds, BD = reference_dataset(training, clip, videos=videos)
    ds.captions_per_video, ds.query_shuffling, ds.caption_length = cpv, shuffling, cap_len
    items = []
    for i in range(len(ds.vid_list)):
      random.seed(1000 + i)
      np.random.seed(2000 + i)
      items.append(ds[i])
    mb = collate(ds, items)
    out[name + '/token_ids'] = mb['token_ids']
    out[name + '/query_masks'] = mb['query_masks']
    if not training:  # (training-mode feature rows are a random choice that follows the caption draws on the SAME generator)
      for e in DF.DIMS:
        out['%s/features_t/%s' % (name, e)] = mb['features_t'][e]
        out['%s/features_ind/%s' % (name, e)] = mb['features_ind'][e]
  np.savez_compressed(OUT_CAP, **out)
  print('wrote', OUT_CAP, os.path.getsize(OUT_CAP), 'bytes')


if __name__ == '__main__':
  main()
