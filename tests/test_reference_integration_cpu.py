"""Reference-side integration (CPU, build container only: skipped where /root/reference is absent).

  * the golden trajectory tests/golden/trainer_epoch.npz IS what the reference's own `Trainer._train_epoch` produces, and
    `trainer_harness.mimic_train_epoch` (what the GPU test drives the MI355X drop-in with) is that method, statement for
    statement: bit-identical losses and weights on the reference model;
  * `BaseTrainer.match_checkpoint_to_model` (base/base_trainer.py:391-406: zero-padding of
    vid_bert.embeddings.position_embeddings.weight, e.g. a 32-position MSRVTT checkpoint into a 102-position model)
    followed by `load_state_dict` works on the drop-in's state dict;
  * the shim of INTEGRATION.md section 1(b) makes `config.init('arch', model.model, ...)` (parse_config.py:138-145,
    train.py:86-91) build the drop-in from a PUBLISHED config's arch arguments, and the trainer's imported
    `sharded_cross_view_inner_product` resolve to ours."""
import importlib
import json
import os
import warnings

import numpy as np
import pytest
import torch

from oracle.ref_loader import REFERENCE_ROOT, load_reference, reference_available
from tests import trainer_harness as H
from tests.fixtures import load_npz

pytestmark = pytest.mark.skipif(not reference_available(), reason='needs the reference tree (build container)')


def test_real_trainer_epoch_reproduces_golden_and_mimic_is_the_same_loop():
  from oracle import gen_trainer_golden as G
  R = load_reference()
  g = load_npz('trainer_epoch')
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model, sd, losses, logs, tr = G.run_real(R)
  assert np.abs(np.asarray(losses) - g['losses']).max() < 1e-6
  assert np.abs(np.asarray([l['loss'] for l in logs]) - g['epoch_loss']).max() < 1e-6
  assert abs(tr.optimizer.param_groups[0]['lr'] - float(g['final_lr'])) < 1e-12
  final = model.state_dict()
  for k in H.PROBE_PARAMS:
    assert np.abs(final[k].reshape(-1)[::37][:4096].numpy() - g['final/' + k]).max() < 1e-6, k
  m2, losses2, logs2, st = G.run_mimic(R)
  assert losses2 == losses
  for (k, a), (_, b) in zip(final.items(), m2.state_dict().items()):
    assert torch.equal(a, b), k
  assert (st.n_samples, st.n_steps) == (tr.n_samples, tr.n_steps)


def test_real_valid_epoch_reproduces_golden_and_mimic_is_the_same_loop():
  """Evaluation half: the reference's own `Trainer._valid_epoch` (+ `_get_embeddings`, trainer/trainer.py:286-483) on the
  validation loader with 3 captions per video and masked captions gives tests/golden/trainer_valid.npz, and
  `trainer_harness.mimic_valid_epoch` (what the GPU test drives the drop-in with) is that method: identical similarity
  matrix, metrics and rank vectors."""
  from oracle import gen_trainer_golden as G
  R = load_reference()
  g = load_npz('trainer_valid')
  model, _ = G.build_reference_model(R)
  sims, qm, metrics = G.run_real_valid(R, model)
  assert np.abs(sims - g['sims']).max() < 1e-6 and np.array_equal(qm, g['query_masks'])
  want = json.loads(str(g['metrics']))
  for name in ('t2v_metrics', 'v2t_metrics'):
    for k in H.METRIC_KEYS:
      assert abs(float(metrics[name][k]) - want[name][k]) < 1e-4, (name, k)
  sims2, nested2, _ = H.mimic_valid_epoch(model, H.MODS, H.EvalLoader(), torch.device('cpu'),
                                          R.model.sharded_cross_view_inner_product,
                                          [R.metric.t2v_metrics, R.metric.v2t_metrics])
  assert np.array_equal(sims, sims2)
  for name in ('t2v_metrics', 'v2t_metrics'):
    assert np.array_equal(metrics[name]['cols'], nested2[name]['cols'])
  # and the oracle's restatement of model/metric.py (what the GPU tests compare the device-side ranks with) agrees
  from oracle import mmt_oracle as O
  for name, fn in (('t2v_metrics', O.t2v_metrics), ('v2t_metrics', O.v2t_metrics)):
    got = fn(sims, query_masks=qm)
    for k in H.METRIC_KEYS:
      assert abs(got[k] - float(metrics[name][k])) < 1e-4, (name, k)


def _native_cenet(max_pos, txt_bert=None):
  from mmt_amd import synthetic
  from mmt_amd.model import CENet
  args = H.arch_args()
  args['vid_bert_params'] = dict(args['vid_bert_params'], max_position_embeddings=max_pos)
  return CENet(expert_dims=synthetic.compute_dims(H.MODS), tokenizer=None, txt_bert=txt_bert or H.HashTextTower(), **args)


def test_match_checkpoint_to_model_then_load_state_dict():
  load_reference()
  BT = importlib.import_module('base.base_trainer').BaseTrainer
  small, big = _native_cenet(32), _native_cenet(102)
  ckpt = {k: v.detach().clone() for k, v in small.state_dict().items()}
  key = 'vid_bert.embeddings.position_embeddings.weight'
  assert ckpt[key].shape[0] == 32 and big.state_dict()[key].shape[0] == 102
  BT.match_checkpoint_to_model(None, ckpt, big.state_dict())        # base/base_trainer.py:391-406
  assert ckpt[key].shape[0] == 102
  big.load_state_dict(ckpt, strict=True)                            # base/base_trainer.py:432
  got = big.state_dict()[key]
  assert torch.equal(got[:32], small.state_dict()[key]) and float(got[32:].abs().max()) == 0.0
  for k, v in small.state_dict().items():
    if k != key:
      assert torch.equal(big.state_dict()[k], v), k


def test_integration_shim_builds_the_drop_in_from_a_published_config(monkeypatch):
  R = load_reference()
  import mmt_amd.loss
  import mmt_amd.model
  T = importlib.import_module('trainer.trainer')
  # INTEGRATION.md section 1(b)
  monkeypatch.setattr(R.model, 'CENet', mmt_amd.model.CENet)
  monkeypatch.setattr(R.model, 'sharded_cross_view_inner_product', mmt_amd.model.sharded_cross_view_inner_product)
  monkeypatch.setattr(R.loss, 'MaxMarginRankingLoss', mmt_amd.loss.MaxMarginRankingLoss)
  monkeypatch.setattr(T, 'sharded_cross_view_inner_product', mmt_amd.model.sharded_cross_view_inner_product)
  cfg = json.load(open(os.path.join(REFERENCE_ROOT, 'configs_pub', 'eccv20', 'MSRVTT_jsfusion_trainval.json')))
  expert_dims = R.util.compute_dims(cfg)
  # parse_config.py:138-145: getattr(module, cfg[name]['type'])(*args, **cfg[name]['args'], **kwargs)
  arch = cfg['arch']
  model = getattr(R.model, arch['type'])(expert_dims=expert_dims, tokenizer=None, txt_bert=H.HashTextTower(),
                                         **arch['args'])
  assert isinstance(model, mmt_amd.model.CENet) and isinstance(model, torch.nn.Module)
  assert list(model.modalities) == list(expert_dims.keys())
  loss = getattr(R.loss, cfg['loss']['type'])(**cfg['loss']['args'])
  assert isinstance(loss, mmt_amd.loss.MaxMarginRankingLoss)
  # every parameter the reference model has, by name and shape (released checkpoints load)
  want = {k: tuple(v.shape) for k, v in _reference_cenet(arch['args'], expert_dims).state_dict().items()}
  got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
  assert got == want


def _reference_cenet(arch_args, expert_dims):
  """The reference's CENet from a PRIVATE copy of model/model.py (the public attribute is monkeypatched in the test)."""
  import importlib.util
  spec = importlib.util.spec_from_file_location('model_model_private', os.path.join(REFERENCE_ROOT, 'model', 'model.py'))
  ref = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(ref)
  ref.TxtBertModel = H.HashTextTower
  return ref.CENet(expert_dims=expert_dims, tokenizer=None, **arch_args)
