"""Where does the time go when minibatches arrive from pinned host memory?  Variants of the feed in front of the same
captured step (bench.py's workload)."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mmt_amd import synthetic
from mmt_amd.feature_store import RaggedFeatures
from mmt_amd.loss import MaxMarginRankingLoss
from mmt_amd.train_step import FlatMinibatch, GraphedTrainStep

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
B, T = 32, 30
MODS = synthetic.MSRVTT_MODALITIES


def make(i, ragged, where, pin):
  mb, text = synthetic.make_batch(1000 + i, B, MODS, T)
  mb['text'] = text.view(-1, 768)
  if ragged:
    rag = RaggedFeatures.from_dense(mb['features'], mb['features_t'], mb['features_ind'], mb['features_maxpool'], experts=MODS, pin_memory=pin)
    mb = {k: v for k, v in mb.items() if not k.startswith('features')}
    mb['features'] = rag
  return FlatMinibatch(mb, where, pin_memory=pin)


def run(ragged, mode, steps=200):
  torch.manual_seed(0)
  model = bench.build_model(pack=True, text_tower='synthetic').to(dev).train()
  host = mode != 'resident'
  batches = [make(i, ragged, 'cpu' if host and mode != 'prefetch_d2d' else dev, host and mode != 'prefetch_d2d') for i in range(8)]
  static = FlatMinibatch(batches[0], dev)
  model.txt_bert.text = static['text']
  runner = GraphedTrainStep(model, MaxMarginRankingLoss(0.05, True), static, lr=5e-5)
  it = 0
  def feed():
    nonlocal it
    if mode in ('resident', 'host_sync'):
      runner.load(batches[it % 8]); it += 1
    else:
      runner.load_prefetched(); it += 1
      runner.prefetch(batches[it % 8])
  if mode.startswith('prefetch'):
    runner.prefetch(batches[0])
  for _ in range(20):
    feed(); runner.step()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    feed(); runner.step()
  t_cpu = time.perf_counter() - t0
  torch.cuda.synchronize()
  el = time.perf_counter() - t0
  print('ragged=%d %-14s %.4f ms/step  (host loop alone %.4f ms/step)' % (ragged, mode, el / steps * 1e3, t_cpu / steps * 1e3), flush=True)


for ragged in (0, 1):
  for mode in ('resident', 'prefetch', 'prefetch_d2d'):
    run(ragged, mode)
