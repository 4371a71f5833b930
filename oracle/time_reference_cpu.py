"""TEST / MEASUREMENT INFRASTRUCTURE (build container only: needs /root/reference) -- times ONE training step of the
UNMODIFIED reference on CPU for BASELINE.md section 4: `model.model.CENet` (train mode, dropout 0.1 as published,
model/model.py:312-661) + `MaxMarginRankingLoss` (model/loss.py:29-65) + backward + `torch.optim.Adam.step`
(train.py:98-100), config B of BASELINE.json (MSRVTT jsfusion shape, batch 32, 7 experts x 30 tokens, d512 L4 H4 I3072),
the text tower replaced by synthetic (B, 768) vectors exactly as in bench.py's headline workload -- and bench.py's
`cpu_baseline` (the oracle port) on the same cores right after, so the two are comparable.

    python -m oracle.time_reference_cpu [--steps 6]
"""
import argparse
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmt_amd import synthetic  # noqa: E402
from oracle import gen_golden as G  # noqa: E402
from oracle.ref_loader import load_reference  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=6)
  args = ap.parse_args()
  R = load_reference()
  threads = min(32, os.cpu_count())
  torch.set_num_threads(threads)
  fx = dict(G.FIXTURES['configB'])
  arch = G.arch_args(fx)
  arch['vid_bert_params'] = synthetic.vid_bert_params(dropout=0.1, **fx['vb'])
  arch['txt_bert_params'] = {'hidden_dropout_prob': 0.1, 'attention_probs_dropout_prob': 0.1}
  orig = G.arch_args
  G.arch_args = lambda _fx: arch
  try:
    model, _ = G.build_reference_cenet(R, fx)
  finally:
    G.arch_args = orig
  model.train()
  loss_fn = R.loss.MaxMarginRankingLoss(margin=0.05, fix_norm=True)
  opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=5e-5)
  mb, text = synthetic.make_batch(0, fx['batch'], fx['modalities'], fx['max_tokens'], max_pos=fx['vb']['max_pos'])
  model.txt_bert.text = text.view(-1, text.shape[-1])
  times = []
  for _ in range(args.steps + 1):
    m = copy.deepcopy(mb)
    t0 = time.time()
    opt.zero_grad()
    out = model(m['token_ids'], m['features'], m['features_t'], m['features_ind'], m['features_avgpool'],
                m['features_maxpool'], m['query_masks'], out='conf', device='cpu')
    loss_fn(out['cross_view_conf_matrix']).backward()
    opt.step()
    times.append(time.time() - t0)
  sec = sum(times[1:]) / args.steps
  with open('/proc/cpuinfo') as f:
    cpu = next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
  ref = dict(kind='reference', value=fx['batch'] / sec, unit='pairs/s', s_per_step=sec, cores=threads, cpu=cpu,
             sample='%d training steps after 1 warm-up' % args.steps)
  print(json.dumps(ref))
  print(json.dumps(bench.cpu_baseline(args.steps)))


if __name__ == '__main__':
  main()
