"""Fused flat Adam (mmt_adam_step) against torch.optim.Adam (the reference's optimizer, train.py:100), including a
learning-rate schedule followed by a CAPTURED optimizer graph (the rate lives in a device scalar)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(seed=0):
  from mmt_amd.flat import FlatParams
  torch.manual_seed(seed)
  dev = torch.device('cuda', 0)
  shapes = [(37, 64), (64,), (128, 96), (5,), (256, 256)]
  params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
  ref = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in params]
  flat = FlatParams([('p%d' % i, p) for i, p in enumerate(params)])
  flat.ensure(dev)
  return dev, params, ref, flat


def _set_grads(flat, params, ref, step):
  g = flat.current_grad()
  gen = torch.Generator(device='cuda').manual_seed(100 + step)
  g.copy_(torch.randn(g.shape, device=g.device, generator=gen) * 0.1)
  for p, r in zip(params, ref):
    r.grad = flat.view(p, g).detach().clone()


@pytest.mark.parametrize('weight_decay', [0.0, 0.01])
def test_flat_adam_matches_torch_adam(weight_decay):
  from mmt_amd.optim import FlatAdam
  dev, params, ref, flat = _setup()
  opt = FlatAdam(flat, lr=1e-3, weight_decay=weight_decay)
  topt = torch.optim.Adam(ref, lr=1e-3, weight_decay=weight_decay)
  sched = torch.optim.lr_scheduler.StepLR(topt, step_size=2, gamma=0.5)
  for step in range(6):
    _set_grads(flat, params, ref, step)
    opt.lr = topt.param_groups[0]['lr']  # same schedule, through the torch-like param_groups interface
    opt.step()
    topt.step()
    sched.step()
  for p, r in zip(params, ref):
    assert torch.allclose(p.detach(), r.detach(), rtol=2e-5, atol=2e-6), (p - r).abs().max().item()


def test_captured_optimizer_graph_follows_the_lr_schedule():
  from mmt_amd.optim import FlatAdam
  dev, params, ref, flat = _setup(1)
  opt = FlatAdam(flat, lr=1e-3)
  topt = torch.optim.Adam(ref, lr=1e-3)
  s = torch.cuda.Stream()
  s.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(s):
    _set_grads(flat, params, ref, 0)
    opt.step()  # eager warm-up step allocates the state
    topt.step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
      opt.step()
    # the capture itself does not execute: state is as after step 1
    for step, lr in enumerate([1e-3, 5e-4, 5e-4, 1e-4], start=1):
      _set_grads(flat, params, ref, step)
      opt.param_groups[0]['lr'] = lr
      topt.param_groups[0]['lr'] = lr
      opt.sync_lr()
      graph.replay()
      topt.step()
  torch.cuda.synchronize()
  for p, r in zip(params, ref):
    assert torch.allclose(p.detach(), r.detach(), rtol=2e-5, atol=2e-6), (p - r).abs().max().item()


def test_fused_adam_refreshes_the_bf16_shadows():
  """mmt_adam_step_fused: same update as torch.optim.Adam AND the bf16 W / W^T shadows equal a re-pack of the updated
  master (ragged 64x64 tile edges, a fused q|k|v block, a zero-padded leading dimension, plain spans in between)."""
  from mmt_amd.flat import FlatParams
  from mmt_amd.optim import FlatAdam
  torch.manual_seed(3)
  dev = torch.device('cuda', 0)
  shapes = [('a', (100, 72)), ('bias_a', (100,)), ('q', (64, 128)), ('k', (64, 128)), ('v', (64, 128)), ('ln', (128,)),
            ('r', (96, 300)), ('tail', (7,))]
  params = [torch.nn.Parameter(torch.randn(*s) * 0.3) for _, s in shapes]
  ref = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in params]
  flat = FlatParams([(n, p) for (n, _), p in zip(shapes, params)])
  P = dict(zip([n for n, _ in shapes], params))
  flat.add_shadow('a', [P['a']], 100, 72, transpose=False)
  flat.add_shadow('qkv', [P['q'], P['k'], P['v']], 192, 128, transpose=True)
  flat.add_shadow('r', [P['r']], 96, 300, k_pad=384)
  flat.ensure(dev)
  flat.pack()
  opt = FlatAdam(flat, lr=1e-2)
  topt = torch.optim.Adam(ref, lr=1e-2)
  for step in range(3):
    _set_grads(flat, params, ref, step)
    opt.step()
    topt.step()
    assert opt._seg_table is not None, 'the fused path must be taken'
    assert not flat._dirty
  for p, r in zip(params, ref):
    assert torch.allclose(p.detach(), r.detach(), rtol=2e-5, atol=2e-6), (p - r).abs().max().item()
  got = {k: tuple(None if t is None else t.clone() for t in flat.shadow(k)) for k in ('a', 'qkv', 'r')}
  flat.pack(force=True)  # reference: the stand-alone packing kernel on the updated master
  for k in got:
    for mine, want in zip(got[k], flat.shadow(k)):
      if want is not None:
        assert torch.equal(mine, want), k
  qkv_t = flat.shadow('qkv')[1]
  want = torch.cat([P['q'], P['k'], P['v']], 0).detach().to(torch.bfloat16).t()
  assert torch.equal(qkv_t[:128, :192], want)
  assert flat.shadow('r')[0][:, 300:].abs().max().item() == 0.0


def test_flat_adam_state_dict_round_trips_through_torch_adam():
  """FlatAdam.state_dict() IS a torch.optim.Adam state dict: load it into a stock Adam over the same parameters, keep
  stepping both, and load the stock optimizer's state back -- weights, moments and step count stay together."""
  from mmt_amd.optim import FlatAdam
  dev, params, ref, flat = _setup(5)
  opt = FlatAdam(flat, lr=1e-3)
  for step in range(3):
    _set_grads(flat, params, ref, step)
    opt.step()
  for p, r in zip(params, ref):
    r.data.copy_(p.data)
  topt = torch.optim.Adam(ref, lr=123.0)           # every hyper-parameter comes from the state dict
  topt.load_state_dict(opt.state_dict())
  assert topt.param_groups[0]['lr'] == 1e-3 and int(topt.state[ref[0]]['step']) == 3
  for step in range(3, 6):
    _set_grads(flat, params, ref, step)
    opt.step()
    topt.step()
  for p, r in zip(params, ref):
    assert torch.allclose(p.detach(), r.detach(), rtol=2e-5, atol=2e-6)
  opt2 = FlatAdam(flat, lr=7.0)
  opt2.load_state_dict(topt.state_dict())          # and back
  assert opt2.lr == 1e-3 and int(opt2.step_dev.item()) == 6 and float(opt2.lr_dev.item()) == pytest.approx(1e-3)
  for step in range(6, 8):
    _set_grads(flat, params, ref, step)
    opt2.step()
    topt.step()
  for p, r in zip(params, ref):
    assert torch.allclose(p.detach(), r.detach(), rtol=2e-5, atol=2e-6)


def test_reference_optimizer_checkpoint_splits_over_the_steps_optimizers():
  """A reference checkpoint holds ONE Adam over filter(requires_grad, model.parameters()) (train.py:95-100); the step
  runs a FlatAdam for the flat buffer + a torch Adam for the rest.  merged_state_dict / load_merged_state_dict translate:
  the merged dict loads into a stock single Adam over the whole model, and a stock dict loads back."""
  from mmt_amd.flat import FlatParams
  from mmt_amd.optim import FlatAdam, load_merged_state_dict, merged_state_dict
  torch.manual_seed(9)
  dev = torch.device('cuda', 0)

  class Net(torch.nn.Module):
    def __init__(self):
      super().__init__()
      self.a = torch.nn.Linear(24, 16)      # outside the flat buffer
      self.b = torch.nn.Linear(16, 32)      # flat
      self.c = torch.nn.Linear(32, 8)       # flat
      self.unused = torch.nn.Linear(4, 4)   # never receives a gradient (the pooler's situation)

  net = Net().to(dev)
  flat = FlatParams([('b.weight', net.b.weight), ('b.bias', net.b.bias), ('c.weight', net.c.weight), ('c.bias', net.c.bias)])
  flat.ensure(dev)
  rest = [net.a.weight, net.a.bias, net.unused.weight, net.unused.bias]
  fo, ro = FlatAdam(flat, lr=2e-3), torch.optim.Adam(rest, lr=2e-3)
  twin = Net().to(dev)
  twin.load_state_dict(net.state_dict())
  single = torch.optim.Adam([p for p in twin.parameters() if p.requires_grad], lr=2e-3)

  def grads(step):
    gen = torch.Generator(device='cuda').manual_seed(500 + step)
    for (n, p), (_, q) in zip(net.named_parameters(), twin.named_parameters()):
      if n.startswith('unused'):
        continue
      g = torch.randn(p.shape, device=dev, generator=gen) * 0.1
      q.grad = g.clone()
      if any(p is fp for fp in flat.params):
        flat.view(p, flat.current_grad()).copy_(g)
      else:
        p.grad = g.clone()

  for step in range(3):
    grads(step)
    fo.step(); ro.step(); single.step()
  merged = merged_state_dict(net, [fo, ro])
  want = single.state_dict()
  assert sorted(merged['state']) == sorted(want['state'])          # no entry for the never-stepped parameters
  for i, st in want['state'].items():
    assert float(merged['state'][i]['step']) == float(st['step'])
    assert torch.allclose(merged['state'][i]['exp_avg'], st['exp_avg'], rtol=1e-4, atol=1e-7)
    assert torch.allclose(merged['state'][i]['exp_avg_sq'], st['exp_avg_sq'], rtol=1e-4, atol=1e-9)
  fresh = torch.optim.Adam([p for p in twin.parameters() if p.requires_grad], lr=9.0)
  fresh.load_state_dict(merged)                                      # the reference's resume path accepts it
  fo2, ro2 = FlatAdam(flat, lr=9.0), torch.optim.Adam(rest, lr=9.0)
  load_merged_state_dict(net, [fo2, ro2], want)                      # and a reference checkpoint loads into the split
  for step in range(3, 5):
    grads(step)
    fo2.step(); ro2.step(); fresh.step()
  for (n, p), (_, q) in zip(net.named_parameters(), twin.named_parameters()):
    assert torch.allclose(p.detach(), q.detach(), rtol=2e-5, atol=2e-6), n


def test_graphed_step_optimizer_checkpoint_on_a_real_cenet():
  """GraphedTrainStep.optimizer_state_dict / load_optimizer_state_dict on a real CENet whose step runs TWO optimizers: the
  FlatAdam of the engine's flat buffer and a captured torch Adam (learning rate in a device scalar) for the parameters
  outside it (txt_pro='lin' text heads).  A checkpoint taken after 2 steps and loaded into a runner that has trained on
  restores the reference-layout state: the learning rate reaches BOTH optimizers (the device scalar included), the
  moments of every parameter are the checkpoint's, and both runners then train in lock-step (base/base_trainer.py:353-365,
  426-432)."""
  from mmt_amd import synthetic
  from mmt_amd.loss import MaxMarginRankingLoss
  from mmt_amd.train_step import FlatMinibatch, GraphedTrainStep
  from tests.test_dp_gpu import BATCH, MODS, TOKENS, _build, _slice_batch
  dev = torch.device('cuda', 0)

  def make(lr):
    torch.manual_seed(0)
    model = _build(dev, txt_pro='lin', dropout=0.0)
    mb, text = synthetic.make_batch(33, BATCH, MODS, TOKENS)
    static = FlatMinibatch(_slice_batch(mb, text, slice(0, BATCH)), dev)
    model.txt_bert.text = static['text']
    return model, GraphedTrainStep(model, MaxMarginRankingLoss(0.05, True), static, lr=lr, warmup_steps=1)

  m1, r1 = make(1e-4)
  assert r1.opt_rest is not None and r1._rest_lr is not None  # (the situation the device-scalar learning rate exists for)
  for _ in range(2):
    r1.step()
  torch.cuda.synchronize()
  sd = r1.optimizer_state_dict()
  weights = {k: v.detach().clone() for k, v in m1.state_dict().items()}
  assert sd['param_groups'][0]['lr'] == pytest.approx(1e-4) and len(sd['param_groups']) == 1
  # a second runner with ANOTHER rate that has already stepped (stale moments everywhere, also for parameters the
  # checkpoint has no entry for): load weights + optimizer state, then both must continue identically
  m2, r2 = make(3e-3)
  for _ in range(3):
    r2.step()
  torch.cuda.synchronize()
  m2.load_state_dict(weights)
  r2.weights_changed()  # (the captured step reads bf16 shadows of the weights: regenerate them)
  r2.load_optimizer_state_dict(sd)
  assert r2.opt_flat.lr == pytest.approx(1e-4) and float(r2.opt_flat.lr_dev.item()) == pytest.approx(1e-4)
  assert float(r2._rest_lr.item()) == pytest.approx(1e-4)
  assert int(r2.opt_flat.step_dev.item()) == 2
  assert torch.equal(r2.opt_flat.exp_avg, r1.opt_flat.exp_avg) and torch.equal(r2.opt_flat.exp_avg_sq, r1.opt_flat.exp_avg_sq)
  l1 = [float(r1.step().item()) for _ in range(3)]
  l2 = [float(r2.step().item()) for _ in range(3)]
  torch.cuda.synchronize()
  assert l1 == l2
  for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
    assert torch.equal(p.detach(), q.detach()), n


def _queue_setup(seed):
  """Two identical flat buffers with shadows (ragged tile edges, a fused q|k|v block, plain spans in between)."""
  from mmt_amd.flat import FlatParams
  from mmt_amd.optim import FlatAdam
  dev = torch.device('cuda', 0)
  shapes = [('a', (100, 72)), ('bias_a', (100,)), ('q', (64, 128)), ('k', (64, 128)), ('v', (64, 128)), ('ln', (128,)),
            ('big', (9000,)), ('r', (96, 300)), ('w', (320, 256)), ('tail', (7,))]
  out = []
  for _ in range(2):
    torch.manual_seed(seed)
    params = [torch.nn.Parameter(torch.randn(*s) * 0.3) for _, s in shapes]
    flat = FlatParams([(n, p) for (n, _), p in zip(shapes, params)])
    P = dict(zip([n for n, _ in shapes], params))
    flat.add_shadow('a', [P['a']], 100, 72, transpose=False)
    flat.add_shadow('qkv', [P['q'], P['k'], P['v']], 192, 128, transpose=True)
    flat.add_shadow('r', [P['r']], 96, 300, k_pad=384)
    flat.add_shadow('w', [P['w']], 320, 256, transpose=True)
    flat.ensure(dev)
    flat.pack()
    out.append((flat, params, FlatAdam(flat, lr=1e-2, weight_decay=0.0)))
  return out


@pytest.mark.parametrize('ridden', ['none', 'some', 'all'])
def test_adam_queue_with_riders_is_bit_identical_to_the_fused_step(ridden):
  """mmt_adam_step_queue (+ rider blocks draining part of the queue first, adam_unit.h) against mmt_adam_step_fused:
  master weights, both moments, every bf16 shadow and the step count BIT for bit over three steps; the queue state is
  zero between steps.  Stages: 'w' and 'big' are final first, then q|k|v, the rest only at the end."""
  import ctypes
  from mmt_amd import _lib, ops
  (fa, pa, oa), (fb, pb, ob) = _queue_setup(11)
  stage = {id(p): s for p, s in zip(pb, [None, None, 1, 1, 1, None, 0, None, 0, None])}
  for step in range(3):
    gen = torch.Generator(device='cuda').manual_seed(500 + step)
    g = torch.randn(fa.current_grad().shape, device='cuda', generator=gen) * 0.1
    fa.current_grad().copy_(g)
    fb.current_grad().copy_(g)
    oa.step()  # reference: ONE fused launch
    if step == 0:
      ob._ensure_state()
      n_stages = ob.build_queue(lambda p: stage[id(p)])
      assert n_stages == 2 and 0 < ob.queue_limit(1) < ob.queue_limit(2) < ob._queue['host'].n_units
      ob.arm_queue(True)
    stages = dict(none=0, some=1, all=2)[ridden]
    if stages:  # rider blocks with no host launch: 3 blocks of 2 x 256 threads claim chunks of the first `stages` stages
      _lib.check(_lib.lib().mmt_adam_rider_probe(ob.queue_ptr(), stages, 3, ops._stream()), 'mmt_adam_rider_probe')
      claimed = ob._queue['state'][:2].tolist()  # (a claim counter may overshoot its stage: fetch-add, never retried)
      assert claimed[0] >= ob.queue_limit(1) and (stages < 2 or claimed[1] >= ob.queue_limit(2) - ob.queue_limit(1))
    ob.step()  # the rest of the queue + the step count
    st = ob._queue['state']
    assert int(st[:_lib.RIDER_STAT0].abs().sum()) == 0 and int(st[_lib.RIDER_STAT0 + 64:].abs().sum()) == 0
    assert torch.equal(fa.master, fb.master)
    assert torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
    assert int(oa.step_dev) == int(ob.step_dev) == step + 1
    for k in ('a', 'qkv', 'r', 'w'):
      for x, y in zip(fa.shadow(k), fb.shadow(k)):
        assert (x is None and y is None) or torch.equal(x, y), k
  n, taken, steps = ob.queue_stats()
  assert steps == 3 and taken == 3 * ob.queue_limit(dict(none=0, some=1, all=2)[ridden])


def test_graphed_step_with_adam_riders_trains_bit_identically():
  """GraphedTrainStep(adam_riders=True): the optimizer's units ride in the GEMM launches of the backward (blocks without a
  tile run Adam on parameters whose gradients are final) -- same losses, weights, moments and bf16 shadows, bit for bit, as
  the step whose optimizer is one launch after the backward (dropout on: same masks), and the riders really took work."""
  from mmt_amd import synthetic
  from mmt_amd.loss import MaxMarginRankingLoss
  from mmt_amd.train_step import FlatMinibatch, GraphedTrainStep
  from tests.test_dp_gpu import BATCH, MODS, TOKENS, _build, _slice_batch
  dev = torch.device('cuda', 0)

  def make(riders):
    torch.manual_seed(0)
    model = _build(dev, txt_pro='gbn', dropout=0.1, layers=3)
    mb, text = synthetic.make_batch(33, BATCH, MODS, TOKENS)
    static = FlatMinibatch(_slice_batch(mb, text, slice(0, BATCH)), dev)
    model.txt_bert.text = static['text']
    return model, GraphedTrainStep(model, MaxMarginRankingLoss(0.05, True), static, lr=1e-3, warmup_steps=2,
                                   adam_riders=riders)

  ma, ra = make(False)
  mb_, rb = make(True)
  assert rb._rider_on and not ra._rider_on
  assert int(ma.vid_bert._seed_dev) == int(mb_.vid_bert._seed_dev)
  for i in range(4):
    la, lb = ra.step(), rb.step()
    assert torch.equal(la, lb), (i, float(la), float(lb))
  torch.cuda.synchronize()
  assert torch.equal(ma._flat.master, mb_._flat.master)
  assert torch.equal(ra.opt_flat.exp_avg, rb.opt_flat.exp_avg) and torch.equal(ra.opt_flat.exp_avg_sq, rb.opt_flat.exp_avg_sq)
  assert int(ra.opt_flat.step_dev) == int(rb.opt_flat.step_dev) == 4
  for sa, sb in zip(ma._flat.shadows, mb_._flat.shadows):
    assert torch.equal(sa['dst'], sb['dst'])
    assert sa['dst_t'] is None or torch.equal(sa['dst_t'], sb['dst_t'])
  n, taken, steps = rb.opt_flat.queue_stats()
  assert steps == 4 and 0 < taken <= 4 * rb.opt_flat.queue_limit(99), (n, taken, steps)
