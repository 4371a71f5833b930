"""world_size-2 gloo test of the data-parallel algebra (mmt_amd/dist.py): gathering embeddings, computing
the GLOBAL-batch similarity + loss on every rank and SUM-reducing parameter gradients must equal the
single-process gradient of the same global batch (the reference's DataParallel semantics,
trainer/trainer.py:134,185-199).  The compute here is the CPU oracle standing in for the GPU kernels."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mmt_oracle as O


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _toy(seed=0):
  g = torch.Generator().manual_seed(seed)
  wv = torch.randn(3, 16, 24, generator=g) * 0.3   # per-expert video projections
  wt = torch.randn(3, 16, 20, generator=g) * 0.3   # per-expert text projections
  wm = torch.randn(3, 20, generator=g) * 0.3       # text MoE weights
  xv = torch.randn(8, 24, generator=g)
  xt = torch.randn(8, 20, generator=g)
  return wv, wt, wm, xv, xt


def _embed(wv, wt, wm, xv, xt):
  vid = O.l2_normalize(torch.einsum('mdk,bk->bmd', wv, xv))
  txt = O.l2_normalize(torch.einsum('mdk,bk->bmd', wt, xt))[:, :, None, :]
  tw = torch.softmax(xt @ wm.t(), -1)[:, None, :]
  vw = torch.full((xv.shape[0], 3), 1.0 / 3)
  return {'vid_embds': vid, 'text_embds': txt, 'vid_weights': vw, 'text_weights': tw}


def _loss(e):
  sims = O.cross_view_inner_product(e['vid_embds'], e['text_embds'], e['vid_weights'], e['text_weights'], 'avg')
  return O.max_margin_ranking_loss(sims, 0.2, True)


def _worker(rank, world, port, out):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from mmt_amd import dist as mdist
  wv, wt, wm, xv, xt = _toy()
  params = [p.clone().requires_grad_(True) for p in (wv, wt, wm)]
  b = xv.shape[0] // world
  sl = slice(rank * b, (rank + 1) * b)
  e = _embed(*params, xv[sl], xt[sl])
  g = mdist.gather_embeddings(e)
  assert g['vid_embds'].shape[0] == xv.shape[0]
  loss = _loss(g)
  loss.backward()
  sync = mdist.GradSync(None, params)
  sync.sync()
  if rank == 0:
    torch.save({'loss': loss.detach(), 'grads': [p.grad for p in params]}, out)
  dist.destroy_process_group()


def test_dp_gather_and_sum_reduce_equal_global_batch(tmp_path):
  out = str(tmp_path / 'dp.pt')
  port = _free_port()
  mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
  got = torch.load(out)
  wv, wt, wm, xv, xt = _toy()
  params = [p.clone().requires_grad_(True) for p in (wv, wt, wm)]
  loss = _loss(_embed(*params, xv, xt))
  loss.backward()
  assert abs(loss.item() - got['loss'].item()) < 1e-6
  for p, g in zip(params, got['grads']):
    assert (p.grad - g).abs().max() < 1e-6


def test_single_process_paths_are_identity():
  from mmt_amd import dist as mdist
  x = torch.randn(4, 3, requires_grad=True)
  assert mdist.all_gather_rows(x) is x
  mdist.GradSync(None, [x]).sync()  # no process group: no-op


def _wire_worker(rank, world, port, out):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from mmt_amd import dist as mdist
  res = {}
  for n in (4096, 4099):  # divisible by the world size (reduced in place) and not (zero-padded wire buffer)
    g = torch.Generator().manual_seed(100 + rank)
    base = torch.randn(n + 64, generator=g)
    for algo in ('allreduce', 'rs_ag'):
      for dt in (None, torch.bfloat16):
        flat = base.clone()
        span = flat[32:32 + n]  # a span in the middle of a flat buffer: its neighbours must stay untouched
        wire = mdist.WireBuffer(dt, algo)
        for _ in range(2):  # the wire buffers are reused from step to step
          span.copy_(base[32:32 + n])
          h, fin = wire.reduce(span, None)
          h.wait()
          fin()
        assert torch.equal(flat[:32], base[:32]) and torch.equal(flat[32 + n:], base[32 + n:])
        res[(n, algo, 'bf16' if dt else 'f32')] = span.clone()
  torch.save(res, '%s.%d' % (out, rank))
  dist.barrier()
  dist.destroy_process_group()


def test_reduce_scatter_all_gather_variant_equals_all_reduce(tmp_path):
  """dist.WireBuffer(algo='rs_ag'): reduce_scatter_tensor + all_gather_into_tensor of a gradient span (what RCCL maps to
  the full xGMI mesh) gives every rank the sums the single all_reduce gives -- fp32 bit for bit on two ranks, also when
  the span length is not a multiple of the world size; in the bf16 wire format both round the same addends."""
  out = str(tmp_path / 'wire')
  mp.spawn(_wire_worker, args=(2, _free_port(), out), nprocs=2, join=True)
  r0, r1 = torch.load(out + '.0'), torch.load(out + '.1')
  for n in (4096, 4099):
    want = sum(torch.randn(n + 64, generator=torch.Generator().manual_seed(100 + r))[32:32 + n] for r in range(2))
    for algo in ('allreduce', 'rs_ag'):
      assert torch.equal(r0[(n, algo, 'f32')], r1[(n, algo, 'f32')])
      assert torch.equal(r0[(n, algo, 'f32')], want), (n, algo)
      assert torch.equal(r0[(n, algo, 'bf16')], r1[(n, algo, 'bf16')])
      assert (r0[(n, algo, 'bf16')] - want).abs().max() < 0.05
    assert torch.equal(r0[(n, 'rs_ag', 'bf16')], r0[(n, 'allreduce', 'bf16')])
