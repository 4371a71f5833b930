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
