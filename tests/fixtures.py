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


def text_bert_shapes(cfg):
  """state_dict shapes of a HuggingFace BertModel with the given config fields (HF key names)."""
  d, i = cfg['hidden_size'], cfg['intermediate_size']
  sh = {'embeddings.word_embeddings.weight': (cfg['vocab_size'], d),
        'embeddings.position_embeddings.weight': (cfg['max_position_embeddings'], d),
        'embeddings.token_type_embeddings.weight': (cfg['type_vocab_size'], d),
        'embeddings.LayerNorm.weight': (d,), 'embeddings.LayerNorm.bias': (d,),
        'pooler.dense.weight': (d, d), 'pooler.dense.bias': (d,)}
  for l in range(cfg['num_hidden_layers']):
    p = 'encoder.layer.%d.' % l
    for n in ('attention.self.query', 'attention.self.key', 'attention.self.value', 'attention.output.dense'):
      sh[p + n + '.weight'], sh[p + n + '.bias'] = (d, d), (d,)
    sh[p + 'intermediate.dense.weight'], sh[p + 'intermediate.dense.bias'] = (i, d), (i,)
    sh[p + 'output.dense.weight'], sh[p + 'output.dense.bias'] = (d, i), (d,)
    for n in ('attention.output.LayerNorm', 'output.LayerNorm'):
      sh[p + n + '.weight'], sh[p + n + '.bias'] = (d,), (d,)
  return sh


def load_text_bert_fixture():
  """-> (gold, cfg, state_dict with 'txt_bert.' prefix, input_ids, attention_mask, probe)."""
  gold = load_npz('text_bert')
  meta = json.loads(str(gold['meta']))
  cfg = meta['cfg']
  sd = synthetic.make_state_dict(meta['seed'], {('txt_bert.' + k): v for k, v in text_bert_shapes(cfg).items()})
  sd['txt_bert.embeddings.word_embeddings.weight'][0].zero_()
  for k, c in meta['param_checksums'].items():
    assert abs(synthetic.checksum(sd[k]) - c) <= 1e-6 * max(1.0, abs(c)), 'generator drift: ' + k
  b, w = meta['shape']
  ids, mask = synthetic.text_token_batch(meta['seed'], b, w, cfg['vocab_size'])
  probe = torch.from_numpy(np.random.RandomState(42).randn(b, cfg['hidden_size']).astype(np.float32))
  return gold, cfg, sd, ids, mask, probe
