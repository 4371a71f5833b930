"""Text heads (texthead.hip / texthead2.hip) against a plain PyTorch fp32 restatement of model/model.py:683-750
(GatedEmbeddingUnit + ContextGating + BatchNorm1d) and :262-283,618 (text MoE): outputs, EVERY parameter gradient,
running statistics, the text gradient, with the on-the-fly MoE dropout replayed through its exported mask."""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _params(M, d, K, seed):
  g = torch.Generator().manual_seed(seed)
  r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
  P = []
  for m in range(M):
    P.append(dict(w1=r(d, K, sc=K ** -0.5), b1=r(d, sc=0.1), w2=r(d, d, sc=d ** -0.5), b2=r(d, sc=0.1),
                  bn_gamma=1.0 + r(d, sc=0.1), bn_beta=r(d, sc=0.1), moe_w=r(1, K, sc=K ** -0.5), moe_b=r(1, sc=0.1),
                  running_mean=r(d, sc=0.1), running_var=(1.0 + 0.3 * torch.rand(d, generator=g)).to(DEV)))
  return P


def _reference(P, text, text_moe, C, use_bn, training):
  """-> embds (B, M, C, d), tw (B, C, M), new running stats"""
  N = text.shape[0]
  embs, logits, stats = [], [], []
  for p in P:
    y = F.linear(text, p['w1'], p['b1'])
    x1 = F.linear(y, p['w2'], p['b2'])
    if use_bn:
      rm, rv = p['running_mean'].clone(), p['running_var'].clone()
      x1 = F.batch_norm(x1, rm, rv, p['bn_gamma'], p['bn_beta'], training, 0.1, 1e-5)
      stats.append((rm, rv))
    o = y * torch.sigmoid(x1)
    embs.append(F.normalize(o, dim=-1))
    logits.append(F.linear(text_moe, p['moe_w'], p['moe_b']))
  e = torch.stack(embs, 1)                                  # (N, M, d)
  e = e.view(N // C, C, len(P), -1).permute(0, 2, 1, 3)     # (B, M, C, d)
  tw = F.normalize(F.softmax(torch.cat(logits, -1), dim=1), p=1, dim=-1).view(N // C, C, len(P))
  return e, tw, stats


def _run_native(P, text, C, use_bn, training, de, dtw, drop_p=0.0, seed=None, want_dtext=True, v1=False):
  from mmt_amd import _lib, ops
  from mmt_amd._lib import MmtTextHeads, MmtTextHeadsOpts, check
  L = _lib.lib()
  N, K = text.shape
  M, d = len(P), P[0]['w1'].shape[0]
  h = MmtTextHeads()
  G = [{k: torch.full_like(v, 7.0) for k, v in p.items() if not k.startswith('running')} for p in P]
  run = [dict(running_mean=p['running_mean'].clone(), running_var=p['running_var'].clone()) for p in P]
  w1_all = torch.cat([p['w1'] for p in P], 0).contiguous()
  for m, p in enumerate(P):
    for k in ('b1', 'w2', 'b2', 'bn_gamma', 'bn_beta', 'moe_w', 'moe_b'):
      getattr(h, k)[m] = p[k].data_ptr()
      getattr(h, 'g_' + k)[m] = G[m][k].data_ptr()
    h.w1[m] = w1_all[m * d:(m + 1) * d].data_ptr()
    h.g_w1[m] = G[m]['w1'].data_ptr()
    h.running_mean[m] = run[m]['running_mean'].data_ptr()
    h.running_var[m] = run[m]['running_var'].data_ptr()
  ws = torch.zeros(L.mmt_text_heads_workspace_floats(N, M, d), device=DEV)
  embds = torch.zeros(N // C, M, C, d, device=DEV)
  tw = torch.zeros(N // C, C, M, device=DEV)
  o = MmtTextHeadsOpts()
  key = torch.zeros(1, dtype=torch.int32, device=DEV)
  nbt = torch.full((M,), 5, dtype=torch.long, device=DEV)
  if drop_p > 0:
    o.moe_drop_key = 0x1234
    o.moe_drop_thr16, o.moe_drop_scale = ops.dropout_params(drop_p)
    o.seed_dev, o.key_dev = seed.data_ptr(), key.data_ptr()
  fast = bool(L.mmt_text_heads_fast(N, M, d, K))
  if fast:
    o.num_batches_tracked = nbt.data_ptr()
  check(L.mmt_text_heads_fwd(ctypes.byref(h), ops._p(text), None, N, C, M, d, K, int(use_bn), int(training), ops._p(ws),
                             ops._p(embds), ops._p(tw), ctypes.byref(o), ops._stream()), 'fwd')
  dtext = torch.full_like(text, 7.0) if want_dtext else None
  dmoe = torch.full_like(text, 7.0) if (want_dtext and drop_p > 0) else None
  o.num_batches_tracked = None
  check(L.mmt_text_heads_bwd(ctypes.byref(h), ops._p(text), None, ops._p(w1_all), N, C, M, d, K, int(use_bn),
                             int(training), ops._p(ws), ops._p(de), ops._p(tw), ops._p(dtw), ops._p(dtext), ops._p(dmoe),
                             ctypes.byref(o), ops._stream()), 'bwd')
  torch.cuda.synchronize()
  if dmoe is not None:
    dtext = dtext + dmoe
  return embds, tw, G, run, dtext, nbt


def _close(name, a, b, atol, rtol=1e-4):
  err = (a - b).abs().max().item()
  assert err <= atol + rtol * b.abs().max().item(), '%s: max abs err %.3e (ref max %.3e)' % (name, err, b.abs().max().item())


@pytest.mark.parametrize('N,C,M,d,K,use_bn,training', [
    (32, 1, 7, 512, 768, True, True),     # config B
    (32, 1, 7, 512, 768, True, False),    # eval statistics
    (6, 1, 3, 256, 768, True, True),      # smoke / tiny fixtures
    (12, 2, 2, 256, 96, False, True),     # txt_pro='gem', two captions per video
    (30, 3, 4, 1024, 1024, True, True),   # widest supported
    (40, 1, 3, 256, 768, True, True),     # more than 32 rows: the one-kernel-per-op path (texthead.hip)
    (64, 2, 2, 256, 768, True, False),
])
def test_text_heads_match_torch(N, C, M, d, K, use_bn, training):
  from mmt_amd import _lib
  fast = bool(_lib.lib().mmt_text_heads_fast(N, M, d, K))
  assert fast == (N <= 32)
  P = _params(M, d, K, seed=N + d)
  g = torch.Generator().manual_seed(99)
  text = torch.randn(N, K, generator=g).to(DEV)
  de = torch.randn(N // C, M, C, d, generator=g).to(DEV)
  dtw = torch.randn(N // C, C, M, generator=g).to(DEV)
  # reference with autograd
  Pr = [{k: v.clone().requires_grad_(not k.startswith('running')) for k, v in p.items()} for p in P]
  tr = text.clone().requires_grad_(True)
  e_ref, tw_ref, stats = _reference(Pr, tr, tr, C, use_bn, training)
  ((e_ref * de).sum() + (tw_ref * dtw).sum()).backward()
  embds, tw, G, run, dtext, nbt = _run_native(P, text, C, use_bn, training, de, dtw)
  _close('text_embds', embds, e_ref.detach(), 2e-6)
  _close('text_weights', tw, tw_ref.detach(), 2e-6)
  for m in range(M):
    for k in ('w1', 'b1', 'w2', 'b2', 'moe_w', 'moe_b') + (('bn_gamma', 'bn_beta') if use_bn else ()):
      _close('g_%s[%d]' % (k, m), G[m][k], Pr[m][k].grad, 2e-5, 2e-4)
    if use_bn and training:
      _close('running_mean', run[m]['running_mean'], stats[m][0], 1e-6)
      _close('running_var', run[m]['running_var'], stats[m][1], 1e-6)
  _close('dtext', dtext, tr.grad, 2e-5, 2e-4)
  assert nbt.tolist() == [5 + int(use_bn and training and fast)] * M


def test_text_heads_on_the_fly_moe_dropout_replays():
  """The MoE branch reads dropout(text) without a dropped copy in memory: same mask forward and backward (key saved on
  the device), equal to the stand-alone dropout kernel's mask for the same (key, seed)."""
  from mmt_amd import _lib, ops
  from mmt_amd._lib import check
  N, C, M, d, K = 32, 1, 7, 512, 768
  P = _params(M, d, K, seed=4)
  g = torch.Generator().manual_seed(5)
  text = torch.randn(N, K, generator=g).to(DEV)
  de = torch.randn(N, M, 1, d, generator=g).to(DEV)
  dtw = torch.randn(N, 1, M, generator=g).to(DEV)
  seed = torch.tensor([77], dtype=torch.int32, device=DEV)
  embds, tw, G, run, dtext, _ = _run_native(P, text, C, True, True, de, dtw, drop_p=0.1, seed=seed)
  # the mask the kernels drew = the stand-alone dropout kernel with the same site key and seed
  thr, scale = ops.dropout_params(0.1)
  dropped = torch.empty_like(text)
  ksave = torch.zeros(1, dtype=torch.int32, device=DEV)
  check(_lib.lib().mmt_dropout_f32(ops._p(text), ops._p(dropped), text.numel(), 0x1234, thr, scale, ops._p(seed),
                                   ops._p(ksave), None, ops._stream()), 'dropout')
  mask = (dropped != 0).float() * scale
  assert 0.85 < (dropped != 0).float().mean().item() < 0.95
  Pr = [{k: v.clone().requires_grad_(not k.startswith('running')) for k, v in p.items()} for p in P]
  tr = text.clone().requires_grad_(True)
  e_ref, tw_ref, _ = _reference(Pr, tr, tr * mask, C, True, True)
  ((e_ref * de).sum() + (tw_ref * dtw).sum()).backward()
  _close('text_weights', tw, tw_ref.detach(), 2e-6)
  for m in range(M):
    _close('g_moe_w', G[m]['moe_w'], Pr[m]['moe_w'].grad, 2e-5, 2e-4)
    _close('g_moe_b', G[m]['moe_b'], Pr[m]['moe_b'].grad, 2e-5, 2e-4)
  _close('dtext (fc + masked MoE branch)', dtext, tr.grad, 2e-5, 2e-4)
