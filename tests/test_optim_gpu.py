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
