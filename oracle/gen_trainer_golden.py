"""TEST INFRASTRUCTURE -- tests/golden/trainer_epoch.npz from the REAL reference trainer.

    python -m oracle.gen_trainer_golden        # build container only (needs /root/reference)

Also writes tests/golden/trainer_valid.npz: the reference's own `Trainer._valid_epoch` (+ `_get_embeddings`,
trainer/trainer.py:286-483) on the reference model with the seed weights over a validation loader with 3 captions per video and masked
captions -- the 72 x 24 similarity matrix, every t2v / v2t metric, the gathered embeddings -- and asserts that
tests/trainer_harness.mimic_valid_epoch reproduces it bit for bit.

Runs the reference's own `Trainer._train_epoch` (trainer/trainer.py:120-249) for EPOCHS epochs over an in-memory loader
(BASELINE.json configs[0] shape: 2 experts, 1 BERT layer, batch 8) with the reference's CENet, MaxMarginRankingLoss,
torch.optim.Adam and StepLR (train.py:86-103) on CPU, records every step's loss and a few trained weights, and asserts
that tests/trainer_harness.mimic_train_epoch -- the restatement the GPU test drives the MI355X drop-in with -- reproduces
the real method BIT FOR BIT on the same model."""
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mmt_amd import synthetic  # noqa: E402
from oracle.ref_loader import load_reference  # noqa: E402
from tests import trainer_harness as H  # noqa: E402

LR, GAMMA = 2e-4, 0.95


def build_reference_model(R):
  R.model.TxtBertModel = H.HashTextTower
  expert_dims = R.util.compute_dims({'experts': {'modalities': H.MODS, 'face_dim': 512}})
  torch.manual_seed(0)
  model = R.model.CENet(expert_dims=expert_dims, tokenizer=None, **H.arch_args())
  shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
  sd = synthetic.make_state_dict(H.SEED, shapes)
  model.load_state_dict(sd)
  return model, sd


def optim_for(model):
  opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=LR)   # train.py:98-100
  return opt, torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=GAMMA)              # train.py:102-103


def run_real(R):
  model, sd = build_reference_model(R)
  loss = H._Recorder(R.loss.MaxMarginRankingLoss(margin=0.05, fix_norm=True))
  opt, sched = optim_for(model)
  tr = H.real_trainer(R, model, loss, opt, sched, H.SyntheticLoader(), torch.device('cpu'))
  import warnings
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')  # the reference calls lr_scheduler.get_lr() outside step()
    logs = H.run_epochs(tr._train_epoch)
  return model, sd, loss.values, logs, tr


def run_mimic(R):
  model, _ = build_reference_model(R)
  loss = H._Recorder(R.loss.MaxMarginRankingLoss(margin=0.05, fix_norm=True))
  opt, sched = optim_for(model)
  st = H.MimicState(model, loss, opt, sched, H.SyntheticLoader(), torch.device('cpu'))
  logs = H.run_epochs(lambda ep: H.mimic_train_epoch(st, ep))
  return model, loss.values, logs, st


def run_real_valid(R, model, loader=None):
  """the REAL `_valid_epoch` -> (sims, {metric: {...}}) ; it returns only the metrics, so the similarity matrix is
  caught where the method hands it to the first metric function."""
  seen = {}

  def t2v_metrics(sims, query_masks=None):
    seen['sims'], seen['query_masks'] = np.array(sims), np.array(query_masks)
    return R.metric.t2v_metrics(sims, query_masks=query_masks)

  tr = H.real_valid_trainer(R, model, loader or H.EvalLoader(), torch.device('cpu'))
  tr.metrics = [t2v_metrics, R.metric.v2t_metrics]
  res = tr._valid_epoch(epoch=1, sets='continuous_eval')
  return seen['sims'], seen['query_masks'], res['metrics'][H._EvalDataset.dataset_name]


def valid_golden(R, model):
  sims, qm, metrics = run_real_valid(R, model)
  sims2, nested2, embds = H.mimic_valid_epoch(model, H.MODS, H.EvalLoader(), torch.device('cpu'),
                                              R.model.sharded_cross_view_inner_product,
                                              [R.metric.t2v_metrics, R.metric.v2t_metrics])
  assert np.array_equal(sims, sims2), 'mimic_valid_epoch is not Trainer._valid_epoch'
  for name in ('t2v_metrics', 'v2t_metrics'):
    for k in H.METRIC_KEYS:
      assert metrics[name][k] == nested2[name][k], (name, k)
    assert np.array_equal(metrics[name]['cols'], nested2[name]['cols'])
  out = dict(sims=sims.astype(np.float32), query_masks=qm.astype(np.float32),
             metrics=json.dumps({n: {k: float(metrics[n][k]) for k in H.METRIC_KEYS} for n in ('t2v_metrics', 'v2t_metrics')}),
             t2v_cols=np.asarray(metrics['t2v_metrics']['cols'], np.float64),
             v2t_cols=np.asarray(metrics['v2t_metrics']['cols'], np.float64),
             vid_weights=embds['vid_weights'].numpy(), text_weights=embds['text_weights'].numpy())
  for mod in H.MODS:
    out['vid_embds/' + mod] = embds['vid_embds'][mod].numpy()
    out['text_embds/' + mod] = embds['text_embds'][mod].numpy()
  # how well separated are the ranks?  (smallest gap between a query's positive and any other video, over valid queries)
  np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'trainer_valid.npz'), **out)
  print('real Trainer._valid_epoch == mimic; sims', sims.shape, 't2v', {k: metrics['t2v_metrics'][k] for k in ('R1', 'R5', 'R10')},
        'v2t', {k: metrics['v2t_metrics'][k] for k in ('R1', 'R5', 'R10')})


def separated_golden(R):
  """tests/golden/trainer_valid_sep.npz: the real `_train_epoch` memorises H.SEP_N pairs, the real `_valid_epoch` ranks
  them: a retrieval problem whose positives sit far above every negative (gap recorded), for EXACT R@K comparisons."""
  model, _ = build_reference_model(R)
  loss = H._Recorder(R.loss.MaxMarginRankingLoss(margin=0.05, fix_norm=True))
  opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=H.SEP_LR)
  sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=1.0)
  tr = H.real_trainer(R, model, loss, opt, sched, H.SepTrainLoader(), torch.device('cpu'))
  import warnings
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    H.run_epochs(tr._train_epoch, epochs=H.SEP_EPOCHS)
  sims, qm, metrics = run_real_valid(R, model, H.SepEvalLoader())
  n = sims.shape[0]
  pos = np.diag(sims)
  off = sims + np.where(np.eye(n) > 0, -np.inf, 0.0)
  gap = float(min((pos - off.max(1)).min(), (pos - off.max(0)).min()))  # rows (t2v) and columns (v2t)
  assert gap > 2e-2, gap
  out = dict(sims=sims.astype(np.float32), losses=np.asarray(loss.values, np.float64), min_gap=np.float64(gap),
             metrics=json.dumps({k: {kk: float(metrics[k][kk]) for kk in H.METRIC_KEYS} for k in ('t2v_metrics', 'v2t_metrics')}))
  np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'trainer_valid_sep.npz'), **out)
  print('separated problem: %d steps, loss %.5f -> %.5f, smallest positive-negative gap %.4f, t2v R1 %.1f v2t R1 %.1f' % (
      len(loss.values), loss.values[0], loss.values[-1], gap, metrics['t2v_metrics']['R1'], metrics['v2t_metrics']['R1']))


def main():
  R = load_reference()
  torch.set_num_threads(os.cpu_count())
  model, sd, losses, logs, tr = run_real(R)
  m2, losses2, logs2, st = run_mimic(R)
  assert losses == losses2, (losses, losses2)
  for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
    assert torch.equal(a, b), k
  assert [l['loss'] for l in logs] == [l['loss'] for l in logs2]
  assert (tr.n_samples, tr.n_steps) == (st.n_samples, st.n_steps) == (H.EPOCHS * H.ITERS * H.BATCH, H.EPOCHS * H.ITERS)
  out = dict(meta=json.dumps(dict(lr=LR, gamma=GAMMA, epochs=H.EPOCHS, iters=H.ITERS, seed=H.SEED,
                                  param_shapes={k: list(v.shape) for k, v in sd.items()},
                                  param_checksums={k: synthetic.checksum(v) for k, v in sd.items() if v.dtype.is_floating_point})),
             losses=np.asarray(losses, np.float64), epoch_loss=np.asarray([l['loss'] for l in logs], np.float64),
             final_lr=np.float64(tr.optimizer.param_groups[0]['lr']))
  final = model.state_dict()
  for k in H.PROBE_PARAMS:
    out['init/' + k] = sd[k].reshape(-1)[::37][:4096].numpy()
    out['final/' + k] = final[k].reshape(-1)[::37][:4096].numpy()
  np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'trainer_epoch.npz'), **out)
  print('real Trainer._train_epoch == mimic; losses', ['%.6f' % l for l in losses])
  valid_golden(R, build_reference_model(R)[0])  # evaluation half, on the SEED weights (identical on both sides by construction)
  separated_golden(R)


if __name__ == '__main__':
  main()
