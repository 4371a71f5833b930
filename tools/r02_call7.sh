cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c7
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_trainer_loop_gpu.py -q -x 2>&1 | tail -30 > $O/pytest_trainer.txt
python - > $O/diag.txt 2>&1 <<'PY'
import json, numpy as np, torch
from tests import trainer_harness as H
from tests.fixtures import load_npz
from tests.test_trainer_loop_gpu import _build, DEV
from mmt_amd.loss import MaxMarginRankingLoss
g = load_npz('trainer_epoch'); meta = json.loads(str(g['meta']))
model, sd = _build(meta)
loss = H._Recorder(MaxMarginRankingLoss(margin=0.05, fix_norm=True))
opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=meta['lr'])
sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=meta['gamma'])
st = H.MimicState(model, loss, opt, sched, H.SyntheticLoader(), DEV)
H.run_epochs(lambda ep: H.mimic_train_epoch(st, ep))
print('got ', ['%.6f' % v for v in loss.values]); print('want', ['%.6f' % v for v in g['losses']])
final = model.state_dict()
for k in H.PROBE_PARAMS:
  w0, w1 = g['init/' + k].astype(np.float64), g['final/' + k].astype(np.float64)
  mine = final[k].detach().cpu().reshape(-1)[::37][:4096].double().numpy()
  moved, err = np.linalg.norm(w1 - w0), np.linalg.norm(mine - w1)
  cos = float((mine - w0) @ (w1 - w0) / (np.linalg.norm(mine - w0) * moved + 1e-30))
  print('%-60s err/moved %.3f cos %.4f' % (k, err / moved, cos))
PY
