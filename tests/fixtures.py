"""Loads tests/golden/*.npz and regenerates the seeded inputs they were made from."""
import collections
import json
import os

import numpy as np
import torch

from mmt_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

Fixture = collections.namedtuple('Fixture', 'name gold meta state_dict batch text cfg')


def load_npz(name):
  return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def load_cenet_fixture(name):
  gold = load_npz('cenet_' + name)
  meta = json.loads(str(gold['meta']))
  fx = meta['fixture']
  shapes = {k: tuple(v) for k, v in meta['param_shapes'].items()}
  sd = synthetic.make_state_dict(fx['seed'], shapes)
  for k, c in meta['param_checksums'].items():
    assert abs(synthetic.checksum(sd[k]) - c) <= 1e-6 * max(1.0, abs(c)), 'generator drift: ' + k
  mb, text = synthetic.make_batch(fx['seed'], fx['batch'], fx['modalities'], fx['max_tokens'],
                                  max_pos=fx['vb']['max_pos'])
  assert abs(synthetic.checksum(text) - meta['text_checksum']) < 1e-6 * max(1.0, abs(meta['text_checksum']))
  mods = meta['modalities']
  cfg = dict(modalities=mods, expert_dims=synthetic.compute_dims(fx['modalities']),
             vid_bert_params=synthetic.vid_bert_params(dropout=0.0, **fx['vb']),
             same_dim=fx['vb']['hidden'], test_caption_mode='indep')
  return Fixture(name, gold, meta, sd, mb, text, cfg)


def subsample(t, limit=4096):
  flat = t.detach().reshape(-1)
  stride = max(1, flat.numel() // limit)
  return flat[::stride][:limit].cpu().float().numpy()
