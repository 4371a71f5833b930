"""The N > 1 training step on real kernels: two processes share ONE MI355X and talk through gloo (which moves CUDA
tensors through the host), driving GraphedTrainStep exactly as bench.py does under torch.distributed.run -- embeddings
all-gather, global-batch similarity + loss, backward through the gather, all-reduce(SUM) of the flat gradient buffer,
captured HIP graphs with the collectives between them.  The 2-rank result must equal a single-process run on the
concatenated batch (txt_pro='gem': no BatchNorm, whose per-rank statistics differ by design from global-batch ones)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
MODS = ['ocr', 'speech', 'vggish']
VB = dict(hidden=256, layers=2, heads=2, inter=512, max_pos=32)
BATCH, TOKENS, STEPS = 8, 5, 2


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _build(dev):
  from mmt_amd import synthetic
  from mmt_amd.model import CENet
  from tests.test_host_cpu import _fake_txt_bert
  vb = synthetic.vid_bert_params(dropout=0.0, **VB)
  model = CENet(l2renorm=False, expert_dims=synthetic.compute_dims(MODS), tokenizer=None, keep_missing_modalities=True,
                test_caption_mode='indep', txt_inp='bertftn', txt_agg='bertftn', txt_wgh='emb', vid_wgh='none',
                vid_cont='bert', vid_inp='both', pos_enc='tint', out_tok='mxp', vid_bert_params=vb, txt_pro='gem',
                same_dim=VB['hidden'], txt_bert_params={'hidden_dropout_prob': 0.0, 'attention_probs_dropout_prob': 0.0},
                txt_bert=_fake_txt_bert(), pack_tokens=True)
  sd = synthetic.make_state_dict(21, {k: tuple(v.shape) for k, v in model.state_dict().items()})
  model.load_state_dict(sd)
  return model.to(dev).train()


def _slice_batch(mb, text, sl):
  out = {}
  for k, v in mb.items():
    out[k] = {kk: vv[sl] for kk, vv in v.items()} if isinstance(v, dict) else v[sl]
  out['text'] = text.view(-1, 768)[sl]
  return out


def _run(rank, world, dev, overlap=None):
  from mmt_amd import synthetic
  from mmt_amd.loss import MaxMarginRankingLoss
  from mmt_amd.train_step import FlatMinibatch, GraphedTrainStep
  model = _build(dev)
  mb, text = synthetic.make_batch(33, BATCH, MODS, TOKENS)
  b = BATCH // world
  static = FlatMinibatch(_slice_batch(mb, text, slice(rank * b, (rank + 1) * b)), dev)
  model.txt_bert.text = static['text']
  runner = GraphedTrainStep(model, MaxMarginRankingLoss(0.05, True), static, lr=1e-4, use_graphs=True, warmup_steps=1,
                            overlap_grad_sync=overlap)
  assert runner.staged == (world > 1 if overlap is None else overlap)
  runner._eager_step()  # one un-captured step: its (all-reduced) gradient buffer is what we compare
  torch.cuda.synchronize()
  grad_after_warmup = model._flat.current_grad().detach().clone().cpu()
  losses = [float(runner.loss.item())]
  for _ in range(STEPS):
    losses.append(float(runner.step().item()))
  torch.cuda.synchronize()
  return grad_after_warmup, losses, model._flat.master.detach().clone().cpu()


def _worker(rank, world, port, out):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.cuda.set_device(0)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  g, losses, master = _run(rank, world, torch.device('cuda', 0))
  torch.save({'grad': g, 'losses': losses, 'master': master}, '%s.%d' % (out, rank))
  dist.barrier()
  dist.destroy_process_group()


def test_two_ranks_equal_single_process_global_batch(tmp_path):
  out = str(tmp_path / 'dp')
  mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
  r0, r1 = torch.load(out + '.0'), torch.load(out + '.1')
  g1, losses1, master1 = _run(0, 1, torch.device('cuda', 0))
  # every rank holds the same (global) loss and, after the all-reduce, the same gradient
  assert max(abs(a - b) for a, b in zip(r0['losses'], r1['losses'])) < 1e-6
  assert (r0['grad'] - r1['grad']).abs().max() < 1e-7
  # ... equal to the single-process gradient of the concatenated batch (up to fp32 summation order)
  scale = g1.abs().max().item()
  assert (r0['grad'] - g1).abs().max() < 2e-3 * scale, ((r0['grad'] - g1).abs().max().item(), scale)
  assert max(abs(a - b) for a, b in zip(r0['losses'], losses1)) < 1e-4
  assert r0['losses'][0] > 0 and all(l == l for l in r0['losses'])
  assert (r0['master'] - r1['master']).abs().max() < 1e-6  # replicas stay in lock-step


def test_staged_backward_equals_single_graph_backward():
  """The stage-by-stage backward (graph B cut where gradient spans become final, so that their all-reduce can start
  early) launches the same kernels as the one-graph backward: identical gradients, losses and weights."""
  dev = torch.device('cuda', 0)
  g0, l0, m0 = _run(0, 1, dev, overlap=False)
  g1, l1, m1 = _run(0, 1, dev, overlap=True)
  assert l0 == l1
  assert torch.equal(g0, g1)
  assert torch.equal(m0, m1)
