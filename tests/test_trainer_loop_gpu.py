"""The MI355X drop-in driven by the reference's training loop (GPU box: no reference tree here, so the loop is
`trainer_harness.mimic_train_epoch`, which tests/test_reference_integration_cpu.py pins BIT FOR BIT to the real
`Trainer._train_epoch`, trainer/trainer.py:120-249): `move_dict_to_device` -> `model(**minibatch, out='conf', device=,
debug=)` -> `loss(sims)` -> `backward` -> `torch.optim.Adam.step` -> `loss.item()`, StepLR per epoch -- exactly as
train.py:86-103 wires them, against the trajectory the REAL trainer + REAL model produced on CPU
(tests/golden/trainer_epoch.npz, oracle/gen_trainer_golden.py)."""
import json

import numpy as np
import pytest
import torch

from tests import trainer_harness as H
from tests.fixtures import load_npz

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _build(meta):
  from mmt_amd import synthetic
  from mmt_amd.model import CENet
  model = CENet(expert_dims=synthetic.compute_dims(H.MODS), tokenizer=None, txt_bert=H.HashTextTower(), **H.arch_args())
  shapes = {k: tuple(v) for k, v in meta['param_shapes'].items()}
  assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == shapes  # the reference's names and shapes
  sd = synthetic.make_state_dict(H.SEED, shapes)
  for k, c in meta['param_checksums'].items():
    assert abs(synthetic.checksum(sd[k]) - c) <= 1e-6 * max(1.0, abs(c)), 'generator drift: ' + k
  model.load_state_dict(sd)
  return model.to(DEV), sd


@pytest.mark.parametrize('optimizer', ['torch.optim.Adam', 'FlatAdam'])
def test_drop_in_follows_the_reference_trainer_trajectory(optimizer):
  from mmt_amd.loss import MaxMarginRankingLoss
  from mmt_amd.optim import FlatAdam
  g = load_npz('trainer_epoch')
  meta = json.loads(str(g['meta']))
  model, sd = _build(meta)
  loss = H._Recorder(MaxMarginRankingLoss(margin=0.05, fix_norm=True))
  if optimizer == 'FlatAdam':  # the fused optimizer of INTEGRATION.md: same param_groups interface, StepLR drives it
    model._prepare(DEV)
    flat_ids = {id(p) for p in model.engine_params()}
    opt = FlatAdam(model._flat, lr=meta['lr'])
    rest = [p for p in model.parameters() if p.requires_grad and id(p) not in flat_ids]
    assert all(p.grad is None for p in rest)  # only the (unused) pooler lives outside the flat buffer

    class _Sched:  # StepLR over param_groups[0]['lr'] (FlatAdam is not a torch Optimizer subclass)
      def step(self):
        opt.param_groups[0]['lr'] *= meta['gamma']

      def get_last_lr(self):
        return [opt.param_groups[0]['lr']]
    sched = _Sched()
  else:
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=meta['lr'])   # train.py:98-100
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=meta['gamma'])                 # train.py:102-103
  st = H.MimicState(model, loss, opt, sched, H.SyntheticLoader(), DEV)
  logs = H.run_epochs(lambda ep: H.mimic_train_epoch(st, ep))
  got, want = np.asarray(loss.values), g['losses']
  assert got.shape == want.shape == (meta['epochs'] * meta['iters'],)
  # first step: same weights, bf16 forward -> the parity tolerance of the similarity path (rel 2e-2)
  assert abs(got[0] - want[0]) <= 2e-2 * want[0], (got[0], want[0])
  # later steps: Adam's first updates are ~ lr * sign(g), so bf16-level gradient noise moves individual weights by up to
  # 2 lr; measured: the losses stay within 0.3 % of the reference's (of the largest one) while falling by 5x
  assert np.abs(got - want).max() <= 0.02 * want.max(), (got, want)
  assert got[3:].mean() < 0.5 * got[:3].mean()
  assert abs(opt.param_groups[0]['lr'] - float(g['final_lr'])) < 1e-12
  assert (st.n_samples, st.n_steps) == (meta['epochs'] * meta['iters'] * H.BATCH, meta['epochs'] * meta['iters'])
  final = model.state_dict()
  for k in H.PROBE_PARAMS:
    w0, w1 = g['init/' + k].astype(np.float64), g['final/' + k].astype(np.float64)
    mine = final[k].detach().cpu().reshape(-1)[::37][:4096].double().numpy()
    moved, err = np.linalg.norm(w1 - w0), np.linalg.norm(mine - w1)
    assert err <= 0.15 * moved, (k, err, moved)  # the update the reference made (measured: 1-7 % of it off, bf16 noise)
    cos = float((mine - w0) @ (w1 - w0) / (np.linalg.norm(mine - w0) * moved + 1e-30))
    assert cos > 0.99, (k, cos)
