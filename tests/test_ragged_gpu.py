"""Ragged wire format on the device: a minibatch handed over as `feature_store.RaggedFeatures`
(compact bf16 X_e, uploaded as M + 1 asynchronous copies of the live prefixes, no cast kernel) gives BIT-identical
similarities, gradients and training trajectories to the same minibatch handed over as the reference's dict of dense
fp32 tensors (data_loader/mix_dataset.py:112-144) -- the device rounds those to the same bf16 values itself."""
import numpy as np
import pytest
import torch

from mmt_amd import feature_store as FS

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)
MODS = ['rgb', 's3d', 'vggish']   # the model's expert order (compute_dims: utils/util.py:154-247)
BATCH, TOKENS = 8, 12


def _model(dropout=0.0, seed=21):
  from mmt_amd import synthetic
  from tests.test_host_cpu import _fake_txt_bert
  from mmt_amd.model import CENet
  vb = synthetic.vid_bert_params(hidden=512, layers=2, heads=4, inter=3072, max_pos=32, dropout=dropout)
  m = CENet(l2renorm=False, expert_dims=synthetic.compute_dims(MODS), tokenizer=None, keep_missing_modalities=True,
            test_caption_mode='indep', txt_inp='bertftn', txt_agg='bertftn', txt_wgh='emb', vid_wgh='none',
            vid_cont='bert', vid_inp='both', pos_enc='tint', out_tok='mxp', vid_bert_params=vb, txt_pro='gbn',
            same_dim=512, txt_bert_params={'hidden_dropout_prob': dropout, 'attention_probs_dropout_prob': dropout},
            txt_bert=_fake_txt_bert(), pack_tokens=True)
  sd = synthetic.make_state_dict(seed, {k: tuple(v.shape) for k, v in m.state_dict().items()})
  m.load_state_dict(sd)
  return m.to(DEV)


def _batch(seed):
  from mmt_amd import synthetic
  mb, text = synthetic.make_batch(seed, BATCH, MODS, TOKENS)
  mb['text'] = text.view(-1, 768)
  return mb


def _ragged_minibatch(mb, pin=True):
  """the dict minibatch with its four video entries replaced by ONE RaggedFeatures (host side)"""
  rag = FS.RaggedFeatures.from_dense(mb['features'], mb['features_t'], mb['features_ind'], mb['features_maxpool'],
                                     experts=MODS, pin_memory=pin)
  out = {k: v for k, v in mb.items() if not k.startswith('features')}
  out['features'] = rag
  return out


def _forward(model, mb, feats):
  model.txt_bert.text = mb['text'].to(DEV)
  if isinstance(feats, FS.RaggedFeatures):
    return model(mb['token_ids'].to(DEV), feats, None, None, None, None, mb['query_masks'].to(DEV))
  d = {k: {e: v.to(DEV) for e, v in mb[k].items()} for k in ('features', 'features_t', 'features_ind', 'features_maxpool')}
  return model(mb['token_ids'].to(DEV), d['features'], d['features_t'], d['features_ind'], None, d['features_maxpool'],
               mb['query_masks'].to(DEV))


def test_forward_and_gradients_bit_identical_to_the_dense_dict():
  mb = _batch(7)
  host = _ragged_minibatch(mb)['features']
  assert host.flat.is_pinned()
  dev = FS.RaggedFeatures(host.layout, DEV).copy_from(host)
  assert dev.live == host.live and host.live_bytes() < 0.75 * host.layout.nbytes   # the padding never crosses PCIe
  results = []
  for feats in (None, dev):
    model = _model().train()
    sims = _forward(model, mb, feats)['cross_view_conf_matrix']
    (sims * torch.linspace(-1, 1, sims.numel(), device=DEV).view_as(sims)).sum().backward()
    torch.cuda.synchronize()
    results.append((sims.detach().clone(), model._flat.current_grad().detach().clone()))
  assert torch.equal(results[0][0], results[1][0])
  assert torch.equal(results[0][1], results[1][1])
  assert results[0][1].abs().max() > 0
  # eval / no_grad door (Trainer._get_embeddings): same thing
  model = _model().eval()
  with torch.no_grad():
    a = _forward(model, mb, None)['cross_view_conf_matrix']
    b = _forward(model, mb, dev)['cross_view_conf_matrix']
  assert torch.equal(a, b)


def test_ragged_input_is_validated():
  mb = _batch(7)
  host = _ragged_minibatch(mb, pin=False)['features']
  model = _model()
  with pytest.raises(RuntimeError):
    _forward(model, mb, host)                                   # host buffer: upload it first
  model.pack_tokens = False
  with pytest.raises(NotImplementedError):
    _forward(model, mb, FS.RaggedFeatures(host.layout, DEV).copy_from(host))
  model.pack_tokens = True
  other = FS.RaggedFeatures(FS.RaggedLayout([('s3d', 1024), ('rgb', 2048), ('vggish', 128)], BATCH, TOKENS), DEV)  # wrong order
  with pytest.raises(ValueError):
    _forward(model, mb, other)


def test_captured_training_steps_from_pinned_ragged_minibatches():
  """GraphedTrainStep with the wire format as its static input: load() = header + live-prefix uploads from pinned host
  memory; three optimisation steps with dropout on == the same steps fed with dense device tensors."""
  from mmt_amd.loss import MaxMarginRankingLoss
  from mmt_amd.train_step import FlatMinibatch, GraphedTrainStep
  mbs = [_batch(40 + i) for i in range(3)]
  traj = []
  for ragged in (False, True):
    torch.manual_seed(0)
    model = _model(dropout=0.1).train()
    batches = [FlatMinibatch(_ragged_minibatch(mb) if ragged else mb, 'cpu', pin_memory=True) for mb in mbs]
    static = FlatMinibatch(batches[0], DEV)
    assert (static.ragged == ['features']) == ragged
    model.txt_bert.text = static['text']
    runner = GraphedTrainStep(model, MaxMarginRankingLoss(0.05, True), static, lr=1e-4, warmup_steps=1)
    losses = []
    for b in batches:
      runner.load(b)
      losses.append(float(runner.step().item()))
    torch.cuda.synchronize()
    traj.append((losses, model._flat.master.detach().clone()))
  assert traj[0][0] == traj[1][0], traj
  assert torch.equal(traj[0][1], traj[1][1])
  assert len(set(traj[0][0])) == 3


def test_store_to_device_end_to_end(tmp_path):
  """memory-mapped store -> collator (pinned wire buffer) -> upload -> embeddings == the dense tensors the reference
  pipeline would have produced from the same store, fed through the dict door."""
  from mmt_amd import synthetic
  dims = {e: synthetic.compute_dims(MODS)[e]['dim'] for e in MODS}
  rng = np.random.RandomState(3)
  nvid = 12
  with FS.FeatureStoreWriter(str(tmp_path / 'store'), dims) as w:
    for v in range(nvid):
      feats = {e: rng.randn(int(rng.randint(0, 2 * TOKENS)), d).astype(np.float32) for e, d in dims.items()}
      w.add('v%d' % v, feats)
  store = FS.FeatureStore(str(tmp_path / 'store'))
  coll = FS.RaggedCollator(store, MODS, BATCH, TOKENS, training=False, pin_memory=True)
  host = coll.collate(list(range(2, 2 + BATCH)))
  dev = FS.RaggedFeatures(host.layout, DEV).copy_from(host)
  feats, ft, fi, fm = host.to_dense()
  mb = _batch(9)
  mb.update(features=feats, features_t=ft, features_ind=fi, features_maxpool=fm)
  model = _model().eval()
  with torch.no_grad():
    a = _forward(model, mb, None)['cross_view_conf_matrix']
    b = _forward(model, mb, dev)['cross_view_conf_matrix']
  assert torch.equal(a, b) and a.abs().max() > 0
