"""mmt_simloss_bwd_small (loss + d loss / d sims + similarity backward + read-out backward in ONE launch, n < 64) against
the one-kernel-per-op chain it replaces (mmt_maxmargin / mmt_infonce -> mmt_sims_bwd -> mmt_readout_bwd), which the
reference fixtures pin (tests/test_cenet_gpu.py::test_similarity_losses_match_reference), and against autograd through
the oracle (model/model.py:789-837, model/loss.py:38-81)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _inputs(n, m, d, seed, zero_row=True):
  rs = np.random.RandomState(seed)
  last = torch.from_numpy(rs.randn(n * m, d).astype(np.float32)) * 2.0      # un-normalised read-out rows
  txt = torch.nn.functional.normalize(torch.from_numpy(rs.randn(n, m, d).astype(np.float32)), dim=-1)
  tw = torch.softmax(torch.from_numpy(rs.randn(n, m).astype(np.float32)), -1)
  vw = torch.full((n, m), 1.0 / m)
  if zero_row and n > 2:
    tw[1] = 0.0  # all-zero mixture weights: the 1e-5 normaliser branch (model.py:816)
  return last.to(DEV), txt.to(DEV), tw.to(DEV), vw.to(DEV)


@pytest.mark.parametrize('n,m,d', [(3, 2, 256), (17, 7, 512), (32, 7, 512), (63, 3, 128), (8, 3, 1024)])
@pytest.mark.parametrize('kind,fix_norm', [(0, True), (0, False), (1, True)])
@pytest.mark.parametrize('compact', [True, False])
def test_fused_loss_and_similarity_backward(n, m, d, kind, fix_norm, compact):
  from mmt_amd import _lib, ops
  from mmt_amd._lib import check
  L = _lib.lib()
  assert n <= L.mmt_simloss_small_max_n()
  last, txt, tw, vw = _inputs(n, m, d, 100 + n)
  bm = n * m
  rows_alloc = bm + 5
  # token-row layout of the read-out: compact = rows 0..bm-1, else scattered rows of a bigger buffer
  agg = torch.arange(bm, device=DEV, dtype=torch.int32) if compact else \
      torch.from_numpy(np.random.RandomState(3).permutation(rows_alloc)[:bm].astype(np.int32)).to(DEV)
  big = torch.zeros(rows_alloc, d, device=DEV)
  big[agg.long()] = last
  vid = torch.empty(bm, d, device=DEV)
  inv = torch.empty(bm, device=DEV)
  check(L.mmt_readout_fwd(ops._p(big), ops._p(agg), bm, d, ops._p(vid), ops._p(inv), ops._stream()), 'readout')
  sims = torch.empty(n, n, device=DEV)
  dots = torch.empty(n, n, m, device=DEV)
  check(L.mmt_sims_fwd(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), n, n, m, d, ops._p(sims), ops._p(dots), ops._stream()), 'sims')
  # ---- chain of separate kernels ----
  loss0 = torch.empty((), device=DEV)
  G = torch.empty(n, n, device=DEV)
  scratch = torch.empty(3 * n, device=DEV)
  if kind == 0:
    check(L.mmt_maxmargin(ops._p(sims), n, 0.05, int(fix_norm), ops._p(scratch), ops._p(loss0), ops._p(G), ops._stream()), 'mm')
  else:
    check(L.mmt_infonce(ops._p(sims), n, ops._p(scratch), ops._p(loss0), ops._p(G), ops._stream()), 'nce')
  dtxt0, dvid0, dtw0, dvw0 = (torch.empty_like(x) for x in (txt, vid.view(n, m, d), tw, vw))
  dots0 = dots.clone()
  check(L.mmt_sims_bwd(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), ops._p(dots0), ops._p(G), n, n, m, d, ops._p(dtxt0),
                       ops._p(dvid0), ops._p(dtw0), ops._p(dvw0), ops._stream()), 'sims_bwd')
  dlast0 = torch.zeros(rows_alloc, d, device=DEV)
  check(L.mmt_readout_bwd(ops._p(vid), ops._p(inv), ops._p(dvid0), ops._p(agg), bm, d, ops._p(dlast0), ops._stream()), 'rb')
  # ---- one launch ----
  loss1 = torch.full((), -1.0, device=DEV)
  dtxt1, dvid1, dtw1, dvw1 = (torch.full_like(x, 9.0) for x in (dtxt0, dvid0, dtw0, dvw0))
  dlast1 = torch.zeros(rows_alloc, d, device=DEV)
  check(L.mmt_simloss_bwd_small(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), ops._p(sims), ops._p(dots), n, m, d, kind,
                                0.05, int(fix_norm), ops._p(loss1), ops._p(dtxt1), ops._p(dvid1), ops._p(dtw1), ops._p(dvw1),
                                ops._p(inv), None if compact else ops._p(agg), ops._p(dlast1), ops._stream()), 'fused')
  torch.cuda.synchronize()
  assert abs(loss1.item() - loss0.item()) <= 1e-6 + 1e-5 * abs(loss0.item())
  for name, a, b in (('dtxt', dtxt1, dtxt0), ('dvid', dvid1, dvid0), ('dtw', dtw1, dtw0), ('dvw', dvw1, dvw0),
                     ('dlast', dlast1, dlast0)):
    tol = 1e-6 + 2e-5 * b.abs().max().item()
    assert (a - b).abs().max().item() <= tol, (name, (a - b).abs().max().item(), b.abs().max().item())
  # the video-side outputs are optional (the training step asks for dlast only)
  loss2 = torch.empty((), device=DEV)
  dlast2 = torch.zeros(rows_alloc, d, device=DEV)
  check(L.mmt_simloss_bwd_small(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), ops._p(sims), ops._p(dots), n, m, d, kind,
                                0.05, int(fix_norm), ops._p(loss2), ops._p(dtxt1), None, ops._p(dtw1), None, ops._p(inv),
                                None if compact else ops._p(agg), ops._p(dlast2), ops._stream()), 'fused')
  assert torch.equal(dlast2, dlast1) and loss2.item() == loss1.item()


def test_fused_path_matches_oracle_autograd():
  from mmt_amd import _lib, ops
  from mmt_amd._lib import check
  from oracle import mmt_oracle as O
  L = _lib.lib()
  n, m, d = 32, 7, 512
  last, txt, tw, vw = _inputs(n, m, d, 7, zero_row=False)
  lv = [x.detach().cpu().clone().requires_grad_(True) for x in (last, txt, tw)]
  vid_ref = torch.nn.functional.normalize(lv[0], dim=-1).view(n, m, d)
  sims_ref = O.cross_view_inner_product(vid_ref, lv[1][:, :, None, :], vw.cpu(), lv[2][:, None, :], 'avg')
  loss_ref = O.max_margin_ranking_loss(sims_ref, 0.05, True)
  loss_ref.backward()
  bm = n * m
  agg = torch.arange(bm, device=DEV, dtype=torch.int32)
  vid = torch.empty(bm, d, device=DEV)
  inv = torch.empty(bm, device=DEV)
  check(L.mmt_readout_fwd(ops._p(last), ops._p(agg), bm, d, ops._p(vid), ops._p(inv), ops._stream()), 'readout')
  sims = torch.empty(n, n, device=DEV)
  dots = torch.empty(n, n, m, device=DEV)
  check(L.mmt_sims_fwd(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), n, n, m, d, ops._p(sims), ops._p(dots), ops._stream()), 'sims')
  loss = torch.empty((), device=DEV)
  dtxt, dtw, dlast = torch.empty_like(txt), torch.empty_like(tw), torch.empty_like(last)
  check(L.mmt_simloss_bwd_small(ops._p(txt), ops._p(vid), ops._p(tw), ops._p(vw), ops._p(sims), ops._p(dots), n, m, d, 0, 0.05, 1,
                                ops._p(loss), ops._p(dtxt), None, ops._p(dtw), None, ops._p(inv), None, ops._p(dlast),
                                ops._stream()), 'fused')
  assert (sims.cpu() - sims_ref.detach()).abs().max() < 1e-5
  assert abs(loss.item() - loss_ref.item()) < 1e-6
  for got, ref in ((dlast, lv[0].grad), (dtxt, lv[1].grad), (dtw, lv[2].grad)):
    assert (got.cpu() - ref).abs().max() <= 1e-6 + 1e-4 * ref.abs().max()
