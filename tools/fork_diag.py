"""Where does a forked step graph (GraphedTrainStep fork bits) leave the serial trajectory?  Small model of
tests/test_dp_gpu.py, one eager + N captured steps, flat master compared region by region after every step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import test_dp_gpu as T
from mmt_amd import synthetic
from mmt_amd.loss import MaxMarginRankingLoss
from mmt_amd.train_step import FlatMinibatch, GraphedTrainStep


def run(fork, steps=4, graphs=True):
  dev = torch.device('cuda', 0)
  torch.manual_seed(0)
  model = T._build(dev, txt_pro='gbn', dropout=0.1, layers=4)
  mb, text = synthetic.make_batch(33, T.BATCH, T.MODS, T.TOKENS)
  static = FlatMinibatch(T._slice_batch(mb, text, slice(0, T.BATCH)), dev)
  model.txt_bert.text = static['text']
  r = GraphedTrainStep(model, MaxMarginRankingLoss(0.05, True), static, lr=1e-4, use_graphs=graphs, warmup_steps=1, fork=fork)
  snaps = []
  for i in range(steps):
    l = r.step()
    torch.cuda.synchronize()
    snaps.append((float(l.item()), model._flat.master.detach().clone(), model._flat.current_grad().detach().clone()))
  return snaps, dict(model.grad_regions())


base, regions = run(0)
for fork in [int(x) for x in sys.argv[1:]] or [1, 4, 16, 32, 64, 21]:
  for graphs in (False, True):
    got, _ = run(fork, graphs=graphs)
    msgs = []
    for i, ((l0, m0, g0), (l1, m1, g1)) in enumerate(zip(base, got)):
      bad = [n for n, (o, c) in regions.items() if not torch.equal(m0[o:o + c], m1[o:o + c])]
      badg = [n for n, (o, c) in regions.items() if not torch.equal(g0[o:o + c], g1[o:o + c])]
      msgs.append('step %d loss %s master %s grad %s' % (i, 'same' if l0 == l1 else '%g vs %g' % (l0, l1), bad or 'same', badg or 'same'))
    print('fork %3d graphs=%d: ' % (fork, graphs) + ' | '.join(msgs), flush=True)
