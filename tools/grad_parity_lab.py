"""Relative L2 error of EVERY parameter gradient of the native CENet against autograd through the CPU oracle
(test infrastructure), per fixture: python tools/grad_parity_lab.py [tiny configA configB]"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd.loss import MaxMarginRankingLoss  # noqa: E402
from oracle import mmt_oracle as O  # noqa: E402
from tests.fixtures import load_cenet_fixture  # noqa: E402
from tests.test_host_cpu import build_native_cenet  # noqa: E402

DEV = 'cuda:0'
for name in (sys.argv[1:] or ['tiny', 'configA', 'configB']):
  fx = load_cenet_fixture(name)
  for pack in (False, True):
    model = build_native_cenet(fx.meta, pack_tokens=pack)
    model.load_state_dict(fx.state_dict)
    model.to(DEV).train()
    mb = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v.to(DEV)) for k, v in fx.batch.items()}
    model.txt_bert.text = fx.text.to(DEV).view(-1, fx.text.shape[-1])
    sims = model(mb['token_ids'], mb['features'], mb['features_t'], mb['features_ind'], mb['features_avgpool'],
                 mb['features_maxpool'], mb['query_masks'], out='conf', device=DEV)['cross_view_conf_matrix']
    R = torch.from_numpy(__import__('numpy').random.RandomState(5).randn(*sims.shape).astype('float32'))
    smooth = os.environ.get('SMOOTH', '1') == '1'  # max-margin is piecewise linear: hinge flips at bf16-level sims
    loss = (sims * R.to(DEV)).sum() if smooth else MaxMarginRankingLoss(0.05, True)(sims)  # differences dominate
    loss.backward()
    P = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v.clone()) for k, v in fx.state_dict.items()}
    ref = O.cenet_forward(P, fx.cfg, copy.deepcopy(fx.batch), fx.text, training=True)['cross_view_conf_matrix']
    lref = (ref * R).sum() if smooth else O.max_margin_ranking_loss(ref, 0.05, True)
    lref.backward()
    rows = []
    for k, p in model.named_parameters():
      gr = P[k].grad if k in P else None
      if p.grad is None or gr is None:
        if (p.grad is None) != (gr is None or float(gr.abs().max()) == 0.0):
          print('  MISMATCH presence', k, p.grad is None, gr is None)
        continue
      g = p.grad.detach().cpu().double()
      r = gr.double()
      rel = float((g - r).norm() / (r.norm() + 1e-30))
      rows.append((rel, k, float(r.norm())))
    rows.sort(reverse=True)
    print('%s pack=%d loss %.6f ref %.6f  params %d  worst rel-L2:' % (name, pack, loss.item(), lref.item(), len(rows)))
    for rel, k, n in rows[:int(os.environ.get("TOPN", "8"))]:
      print('   %.3e  |ref| %.3e  %s' % (rel, n, k))
    print('   median %.3e' % rows[len(rows) // 2][0])
