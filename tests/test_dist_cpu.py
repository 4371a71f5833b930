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


# ---- world size 8 (the size the driver's multi-GPU run uses): collective geometry on CPU / gloo -------------------------
WORLD8 = 8


def _wire8_worker(rank, world, port, out):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from mmt_amd import dist as mdist
  res = {}
  # lengths: a multiple of 4 * world (reduced in place), not a multiple (zero-padded wire buffer), shorter than 4 * world
  # (the last ranks own NOTHING of the span), and a single element
  for n in (4096, 4099, 37, 5, 1):
    g = torch.Generator().manual_seed(100 + rank)
    base = torch.randn(n + 64, generator=g)
    for algo in ('allreduce', 'rs_ag'):
      for dt in (None, torch.bfloat16):
        flat = base.clone()
        span = flat[32:32 + n]
        wire = mdist.WireBuffer(dt, algo)
        for _ in range(2):
          span.copy_(base[32:32 + n])
          h, fin = wire.reduce(span, None)
          h.wait()
          fin()
        assert torch.equal(flat[:32], base[:32]) and torch.equal(flat[32 + n:], base[32 + n:]), (n, algo, dt)
        res[(n, algo, 'bf16' if dt else 'f32')] = span.clone()
    # the halves on their own, as the sharded optimizer uses them
    wire = mdist.WireBuffer(None, 'rs_ag')
    flat = base.clone()
    span = flat[32:32 + n]
    work, shard, _, lo, own = wire.reduce_scatter(span, None)
    work.wait()
    per = wire._shard_geometry(n, world)
    assert lo == rank * per and per % 4 == 0 and own == max(0, min(per, n - lo)) and per * world >= n
    res[(n, 'shard')] = (lo, own, wire.shard_f32(span, shard)[:own].clone())
  torch.save(res, '%s.%d' % (out, rank))
  dist.barrier()
  dist.destroy_process_group()


def test_wire_buffer_geometry_at_world_size_eight(tmp_path):
  """Every exchange of dist.WireBuffer on EIGHT ranks: all-reduce and reduce-scatter + all-gather, fp32 and bf16 wire,
  span lengths that are / are not multiples of 4 x 8 elements, spans so short that the last ranks own nothing.  The shards
  tile the span exactly once and hold the global sums; neighbours of the span in the flat buffer are never touched."""
  out = str(tmp_path / 'wire8')
  mp.spawn(_wire8_worker, args=(WORLD8, _free_port(), out), nprocs=WORLD8, join=True)
  r = [torch.load('%s.%d' % (out, i)) for i in range(WORLD8)]
  for n in (4096, 4099, 37, 5, 1):
    want = sum(torch.randn(n + 64, generator=torch.Generator().manual_seed(100 + k))[32:32 + n].double() for k in range(WORLD8))
    for algo in ('allreduce', 'rs_ag'):
      for k in range(1, WORLD8):
        assert torch.equal(r[0][(n, algo, 'f32')], r[k][(n, algo, 'f32')]), (n, algo, k)
        assert torch.equal(r[0][(n, algo, 'bf16')], r[k][(n, algo, 'bf16')]), (n, algo, k)
      assert (r[0][(n, algo, 'f32')].double() - want).abs().max() < 1e-5, (n, algo)
      assert (r[0][(n, algo, 'bf16')].double() - want).abs().max() < 0.25, (n, algo)
    covered = torch.zeros(n, dtype=torch.int32)
    got = torch.zeros(n, dtype=torch.float64)
    for k in range(WORLD8):
      lo, own, vals = r[k][(n, 'shard')]
      assert vals.numel() == own
      covered[lo:lo + own] += 1
      got[lo:lo + own] = vals.double()
    assert bool((covered == 1).all()), n
    assert (got - want).abs().max() < 1e-5


def _adam_ref(w, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
  """torch.optim.Adam's update on slices (in place)."""
  m.mul_(b1).add_(g, alpha=1 - b1)
  v.mul_(b2).addcmul_(g, g, value=1 - b2)
  denom = (v / (1 - b2 ** step)).sqrt_().add_(eps)
  w.addcdiv_(m / (1 - b1 ** step), denom, value=-lr)


def _shard_adam_worker(rank, world, port, out):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from mmt_amd import dist as mdist
  spans = [(0, 1024), (1024, 4160), (5184, 64), (5248, 192)]  # (offset, count): multiples of flat.ALIGN like grad_regions'
  count = 5440
  w0 = torch.randn(count, generator=torch.Generator().manual_seed(5))
  master, m, v = w0.clone(), torch.zeros(count), torch.zeros(count)
  wire = mdist.WireBuffer(None, 'rs_ag')
  for step in (1, 2, 3):
    grad = torch.randn(count, generator=torch.Generator().manual_seed(1000 * step + rank))
    handles = []
    for off, cnt in spans:  # train_step.GraphedTrainStep._reduce_async in shard mode
      span = grad[off:off + cnt]
      work, shard, _, lo, own = wire.reduce_scatter(span, None)
      handles.append((work, off, cnt, span, shard, lo, own))
    for h in handles:
      h[0].wait()
    for _, off, cnt, span, shard, lo, own in handles:  # ... ._finish: Adam on the rank's shard of every span
      g = wire.shard_f32(span, shard)
      assert (off + lo) % 4 == 0 and own % 4 == 0  # what FlatAdam.step_shard demands of its arguments
      if own:
        a, b = off + lo, off + lo + own
        _adam_ref(master[a:b], g[:own], m[a:b], v[a:b], step)
    gathers = [wire.all_gather_span(master[off:off + cnt], None) for off, cnt in spans]
    for wk, _ in gathers:
      wk.wait()
    for _, fin in gathers:
      fin()
  for off, cnt in spans:  # GraphedTrainStep._gather_sharded_moments: what a checkpoint needs
    for buf in (m, v):
      wk, fin = wire.all_gather_span(buf[off:off + cnt], None, async_op=False)
      if wk is not None:
        wk.wait()
      fin()
  torch.save(dict(master=master, m=m, v=v), '%s.%d' % (out, rank))
  dist.barrier()
  dist.destroy_process_group()


def test_sharded_adam_exchange_at_world_size_eight(tmp_path):
  """The sharded-optimizer exchange of GraphedTrainStep (reduce-scatter of every gradient span, Adam on the rank's shard,
  all-gather of the updated weights, all-gather of the moments for a checkpoint) with EIGHT ranks and spans whose length
  is not a multiple of 4 x 8 -- one of them shorter than the ranks' share, so trailing ranks own nothing: every rank ends
  with the weights AND moments of plain Adam on the summed gradients."""
  out = str(tmp_path / 'sadam8')
  mp.spawn(_shard_adam_worker, args=(WORLD8, _free_port(), out), nprocs=WORLD8, join=True)
  r = [torch.load('%s.%d' % (out, i)) for i in range(WORLD8)]
  count = 5440
  w = torch.randn(count, generator=torch.Generator().manual_seed(5))
  m, v = torch.zeros(count), torch.zeros(count)
  for step in (1, 2, 3):
    g = sum(torch.randn(count, generator=torch.Generator().manual_seed(1000 * step + k)) for k in range(WORLD8))
    _adam_ref(w, g, m, v, step)
  for k in range(WORLD8):
    assert torch.equal(r[0]['master'], r[k]['master']) and torch.equal(r[0]['m'], r[k]['m']) and torch.equal(r[0]['v'], r[k]['v'])
  assert (r[0]['master'] - w).abs().max() < 1e-5
  assert (r[0]['m'] - m).abs().max() < 1e-5 and (r[0]['v'] - v).abs().max() < 1e-4


def test_grad_regions_tile_the_flat_buffer_and_shard_over_eight_ranks():
  """CENet.grad_regions (the spans the staged backward reduces, with and without the split bottom span): disjoint,
  contiguous, in backward order, together the whole flat gradient buffer; cut eight ways by WireBuffer's geometry every
  shard starts on a 16-byte boundary inside its span and the shards tile it."""
  import json
  from tests.fixtures import load_npz
  from tests.test_host_cpu import build_native_cenet
  from mmt_amd import dist as mdist
  meta = json.loads(str(load_npz('cenet_configB')['meta']))
  model = build_native_cenet(meta)
  f = model._flat
  wire = mdist.WireBuffer(None, 'rs_ag')
  for split in (False, True):
    regions = model.grad_regions(split_bottom=split)
    names = [n for n, _ in regions]
    assert names[0] == 'top' and names[-1] == 'bottom' and ('layer0' in names) == split
    spans = sorted(s for _, s in regions)
    assert spans[0][0] == 0 and spans[-1][0] + spans[-1][1] == f.count
    for (o0, c0), (o1, _) in zip(spans, spans[1:]):
      assert o0 + c0 == o1  # adjacent: no gap, no overlap
    # backward order = descending offsets (the flat layout is the reverse of the order gradients become final)
    offs = [s[0] for _, s in regions]
    assert offs == sorted(offs, reverse=True)
    for _, (off, cnt) in regions:
      assert off % 4 == 0 and cnt % 4 == 0
      per = wire._shard_geometry(cnt, WORLD8)
      assert per % 4 == 0 and per * WORLD8 >= cnt and per * (WORLD8 - 1) < cnt + per
      owned = [max(0, min(per, cnt - k * per)) for k in range(WORLD8)]
      assert sum(owned) == cnt and all(o % 4 == 0 for o in owned)
