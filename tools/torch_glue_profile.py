"""Which ATen kernels (torch-side glue around the native calls) does one eager training step launch?
   python tools/torch_glue_profile.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmt_amd import synthetic  # noqa: E402
from mmt_amd.loss import MaxMarginRankingLoss  # noqa: E402
from mmt_amd.train_step import FlatMinibatch, GraphedTrainStep  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bench.build_model(pack=True).to(dev).train()
mb, text = synthetic.make_batch(1000, 32, synthetic.MSRVTT_MODALITIES, 30)
mb['text'] = text.view(-1, 768)
static = FlatMinibatch(mb, dev)
model.txt_bert.text = static['text']
runner = GraphedTrainStep(model, MaxMarginRankingLoss(0.05, True), static, lr=5e-5, use_graphs=False, warmup_steps=3)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
  runner.eager_step()
  torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
  if e.key.startswith('aten::') and e.device_time_total > 0:
    rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:90]))
for t, c, k, sh in sorted(rows, reverse=True)[:40]:
  print('%8.1f us  x%-3d %-28s %s' % (t, c, k, sh))
