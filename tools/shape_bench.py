"""The training step on the other BASELINE.json shapes (not bench lines -- parity-test cases -- but worth a data point):
config 4 = long sequences (7 experts x 100 tokens -> S = 708, max_pos 102), config 5 = the HowTo100M-scale encoder
(d = 1024, L = 6, H = 8, I = 6144).  Same graphed step as bench.py, synthetic inputs resident in HBM."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmt_amd import synthetic  # noqa: E402
from mmt_amd.loss import MaxMarginRankingLoss  # noqa: E402
from mmt_amd.model import CENet  # noqa: E402
from mmt_amd.train_step import FlatMinibatch, GraphedTrainStep  # noqa: E402


def run(name, batch, tokens, hidden, layers, heads, inter, max_pos, steps=60):
  dev = torch.device('cuda', 0)
  vb = synthetic.vid_bert_params(hidden=hidden, layers=layers, heads=heads, inter=inter, max_pos=max_pos, dropout=0.1)
  model = CENet(l2renorm=False, expert_dims=synthetic.compute_dims(synthetic.MSRVTT_MODALITIES), tokenizer=None,
                keep_missing_modalities=True, test_caption_mode='indep', txt_inp='bertftn', txt_agg='bertftn',
                txt_wgh='emb', vid_wgh='none', vid_cont='bert', vid_inp='both', pos_enc='tint', out_tok='mxp',
                vid_bert_params=vb, txt_pro='gbn', same_dim=hidden,
                txt_bert_params={'hidden_dropout_prob': 0.1, 'attention_probs_dropout_prob': 0.1},
                txt_bert=bench.SyntheticTextTower(), pack_tokens=True).to(dev).train()
  batches = []
  for i in range(4):
    mb, text = synthetic.make_batch(2000 + i, batch, synthetic.MSRVTT_MODALITIES, tokens, max_pos=max_pos)
    mb['text'] = text.view(-1, 768)
    batches.append(FlatMinibatch(mb, dev))
  static = FlatMinibatch(batches[0], dev)
  model.txt_bert.text = static['text']
  runner = GraphedTrainStep(model, MaxMarginRankingLoss(0.05, True), static, lr=5e-5)
  for i in range(10):
    runner.load(batches[i % 4]); runner.step()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(steps):
    runner.load(batches[i % 4]); loss = runner.step()
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / steps * 1e3
  seq = 1 + 7 * (tokens + 1)
  flops = 3.0 * batch * seq * layers * (8 * hidden * hidden + 4 * hidden * inter + 4 * seq * hidden)
  plan = model._plans[next(iter(model._plans))]
  print('%-28s B %3d S %4d d %4d L %d | %7.3f ms/step  %8.1f pairs/s  %6.1f TFLOP/s (dense-token accounting)  live rows %d / %d  loss %.4f'
        % (name, batch, seq, hidden, layers, ms, batch / ms * 1e3, flops / ms / 1e9, int(plan.n_rows.item()), batch * seq,
           float(loss.item())))


run('config 2 (MSRVTT, headline)', 32, 30, 512, 4, 4, 3072, 32)
run('config 4 (long sequences)', 32, 100, 512, 4, 4, 3072, 102)
run('config 5 encoder (d1024 L6)', 32, 30, 1024, 6, 8, 6144, 32)
run('config 5 encoder, batch 128', 128, 30, 1024, 6, 8, 6144, 32)
