"""TEST INFRASTRUCTURE -- tests/golden/trainer_epoch.npz from the REAL reference trainer.

    python -m oracle.gen_trainer_golden        # build container only (needs /root/reference)

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


if __name__ == '__main__':
  main()
