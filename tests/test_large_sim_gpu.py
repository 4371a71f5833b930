"""BASELINE.json configs[4] at its OWN size: one rank's row block of the 64k-pair similarity + max-margin loss
(b = 8192 texts x n = 65536 videos, M = 7 experts, d = 1024; mmt_amd/large_sim.py, largesim.hip) checked value by value
on SAMPLED rows and columns against the CPU oracle (model/model.py:789-837, model/loss.py:38-65 restricted to those
rows: oracle.cross_view_rows / max_margin_rows, pinned to the full-matrix oracle in tests/test_oracle_golden.py).
The reference cannot run at this size (its loss materialises 6 vectors of 2 n^2 entries); 16 rows x 65536 columns of it
take the oracle seconds.  Index-width and contention bugs only show here: the block holds 2^29 similarities."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _inputs(n, m, d, seed=3):
  g = torch.Generator(device=DEV).manual_seed(seed)
  nrm = lambda x: torch.nn.functional.normalize(x, dim=-1)
  vid = nrm(torch.randn(n, m, d, device=DEV, generator=g))
  txt = nrm(torch.randn(n, m, d, device=DEV, generator=g) + 0.3 * vid)  # positives correlate: hinges of both kinds
  tw = torch.softmax(torch.randn(n, m, device=DEV, generator=g), -1)
  vw = torch.full((n, m), 1.0 / m, device=DEV)
  return vid, txt, tw, vw


@pytest.mark.parametrize('b,n,rank', [(8192, 65536, 3), (2048, 16384, 7)])
def test_row_block_at_full_size_matches_oracle_on_sampled_rows(b, n, rank):
  from mmt_amd.large_sim import RowBlock
  from oracle import mmt_oracle as O
  m, d, margin = 7, 1024, 0.05
  free, _ = torch.cuda.mem_get_info()
  if free < 48 * 2 ** 30 * (b * n) / (8192 * 65536) + 8 * 2 ** 30:
    pytest.skip('not enough free HBM for the %d x %d row block' % (b, n))
  vid, txt_all, tw_all, vw = _inputs(n, m, d)
  r0 = rank * b
  # the global diagonal s_jj (what the ranks all-gather): O(n M d), in fp64 on the device
  w = tw_all.double() * vw.double()
  w = w / w.sum(-1, keepdim=True)
  diag_all = (w * (txt_all.double() * vid.double()).sum(-1)).sum(-1).float()
  blk = RowBlock(txt_all[r0:r0 + b], tw_all[r0:r0 + b], vid, vw, r0, margin)
  diag_local = blk.phase_similarity()
  assert (diag_local - diag_all[r0:r0 + b]).abs().max().item() < 2e-3
  colcnt, loss = blk.phase_counts(diag_all)
  dtxt, dtw, q = blk.phase_backward(colcnt)  # (single block: the other ranks' column counts are not part of this check)
  torch.cuda.synchronize()

  rs = np.random.RandomState(11)
  rows = np.unique(np.concatenate([[0, b - 1], rs.randint(0, b, size=14)]))       # block-local text rows
  cols = np.unique(np.concatenate([[0, n - 1, r0, r0 + b - 1], rs.randint(0, n, size=12)]))  # global video columns
  R = torch.from_numpy(r0 + rows)
  vid_c, vw_c, diag_c = vid.cpu(), vw.cpu(), diag_all.cpu()
  txt_r = txt_all[r0 + rows].cpu().requires_grad_(True)
  tw_r = tw_all[r0 + rows].cpu().requires_grad_(True)
  s_rows = O.cross_view_rows(txt_r, tw_r, vid_c, vw_c)                              # (16, n)
  got_rows = blk.similarity()[rows].cpu()
  assert (got_rows - s_rows.detach()).abs().max().item() < 2e-3                    # bf16 operands, K = M*d = 7168

  # column hinge counts of the sampled columns over the block's rows: a full column of the block each
  txt_blk, tw_blk = txt_all[r0:r0 + b], tw_all[r0:r0 + b]
  want_cc, band_cc = [], []
  for c in cols:
    wc = tw_blk.double() * vw[c].double()[None, :]
    wc = wc / wc.sum(-1, keepdim=True)
    s_col = (wc * (txt_blk.double() * vid[c].double()[None]).sum(-1)).sum(-1).cpu()   # s[r', c] for every block row
    a = margin - diag_c[c].double() + s_col
    keep = torch.ones(b, dtype=torch.bool)
    if r0 <= c < r0 + b:
      keep[c - r0] = False
    want_cc.append(int(((a > 0) & keep).sum()))
    band_cc.append(int(((a.abs() < 4e-3) & keep).sum()))                            # entries a 2e-3 error can flip
  got_cc = colcnt[cols].cpu().numpy()
  for c, g_, w_, bd in zip(cols, got_cc, want_cc, band_cc):
    assert abs(int(g_) - w_) <= bd, ('colcnt', int(c), int(g_), w_, bd)

  # loss partials, row hinge counts, gradient rows
  cc_rows = colcnt[r0 + rows].cpu()                                                # this block's count for column R
  lp, g_rows, rowcnt = O.max_margin_rows(s_rows.detach(), R, diag_c, margin, n, cc_rows)
  a_row = margin - diag_c[R][:, None] + s_rows.detach()
  band = ((a_row.abs() < 4e-3).sum(1) + ((margin - diag_c[None, :] + s_rows.detach()).abs() < 4e-3).sum(1))
  got_rc = blk.rowcnt[rows].cpu()
  assert ((got_rc - rowcnt).abs() <= band).all(), (got_rc, rowcnt, band)
  got_lp = (blk.loss_part[rows] / blk.norm).cpu()
  assert ((got_lp - lp).abs() <= 2e-2 * lp.abs() + 1e-9).all(), (got_lp, lp)
  (s_rows * g_rows).sum().backward()                                                # d loss / d (T_R, tw_R) of the sampled rows
  for name, got, want in (('dtxt', dtxt[rows].cpu(), txt_r.grad), ('dtw', dtw[rows].cpu(), tw_r.grad)):
    gv, wv = got.double().reshape(-1), want.double().reshape(-1)
    cos = float(gv @ wv / (gv.norm() * wv.norm()))
    assert cos > 0.99 and abs(float(gv.norm() / wv.norm()) - 1.0) < 0.05, (name, cos, float(gv.norm() / wv.norm()))
  # the whole block: finite, |S| <= 1 for unit-norm inputs, counts within range
  assert torch.isfinite(loss) and torch.isfinite(dtxt).all() and torch.isfinite(q).all()
  assert blk.similarity().abs().max().item() <= 1.0 + 2e-3
  assert int(blk.rowcnt.max()) < n and int(colcnt.max()) <= b


@pytest.mark.parametrize('m', [7, 12])
def test_sweep_variants_agree_bit_for_bit(m):
  """The sweeps in their three forms -- (a) finish in place, then counts / gradient on the finished block with vw as [n, M]
  (mmt_ls_finish, mmt_ls_counts, mmt_ls_grad); (b) one finishing sweep (finish = 1) + gradient; (c) S left raw, both passes
  divide on the fly and read vw transposed (finish = 2, raw = 1, vw_t) -- produce identical hinge counts and G' (the division
  is the same two instructions everywhere: largesim.hip ls_quot) and loss / gs partial sums equal up to fp32 association.  Video weights vary per column here (zeros included: the 1e-5 branch)."""
  from mmt_amd import _lib, ops
  from mmt_amd._lib import check
  L = _lib.lib()
  b, n, r0, margin = 43, 2048 + 512, 256, 0.05  # (43: a partial last row block -- the checked copy of the sweeps)
  g = torch.Generator(device=DEV).manual_seed(5)
  raw = torch.randn(b, n, device=DEV, generator=g) * 0.05
  tw = torch.softmax(torch.randn(b, m, device=DEV, generator=g), -1)
  vw = torch.softmax(torch.randn(n, m, device=DEV, generator=g), -1)
  vw[5] = 0.0                                                    # den == 0 -> 1e-5, no normaliser gradient
  vw_t = vw.t().contiguous()
  diag = torch.randn(n, device=DEV, generator=g) * 0.05
  ncb = L.mmt_ls_col_blocks(n)
  inv_norm = 1.0 / (2.0 * n * (n - 1))

  def run(kind):
    S = raw.clone()
    rowcnt = torch.zeros(b, device=DEV, dtype=torch.int32)
    colcnt = torch.zeros(n, device=DEV, dtype=torch.int32)
    part = torch.zeros(b, ncb, device=DEV)
    g16 = torch.zeros(b, n, device=DEV, dtype=torch.bfloat16)
    gs = torch.zeros(b, ncb, m, device=DEV)
    st = ops._stream()
    if kind == 'a':
      check(L.mmt_ls_finish(ops._p(S), n, ops._p(tw), ops._p(vw), b, n, m, st), 'finish')
      check(L.mmt_ls_counts(ops._p(S), n, ops._p(diag), b, n, r0, margin, ops._p(rowcnt), ops._p(colcnt), ops._p(part), st), 'counts')
      check(L.mmt_ls_grad(ops._p(S), n, ops._p(diag), ops._p(tw), ops._p(vw), ops._p(rowcnt), ops._p(colcnt), b, n, m, r0, margin,
                          inv_norm, ops._p(g16), n, ops._p(gs), st), 'grad')
    else:
      fin, vt = (1, None) if kind == 'b' else (2, ops._p(vw_t))
      check(L.mmt_ls_counts_ex(ops._p(S), n, ops._p(diag), ops._p(tw), ops._p(vw), vt, m, fin, b, n, r0, margin, ops._p(rowcnt),
                               ops._p(colcnt), ops._p(part), st), 'counts_ex')
      check(L.mmt_ls_grad_ex(ops._p(S), n, ops._p(diag), ops._p(tw), ops._p(vw), vt, ops._p(rowcnt), ops._p(colcnt), b, n, m, r0,
                             margin, inv_norm, ops._p(g16), n, ops._p(gs), 1 if kind == 'c' else 0, st), 'grad_ex')
      if kind == 'c':
        assert torch.equal(S, raw)                               # the row block was never rewritten
    return rowcnt, colcnt, part, g16, gs

  ref = run('a')
  assert int(ref[0].sum()) > 0 and int(ref[1].sum()) > 0
  for kind in ('b', 'c'):
    for name, x, y in zip(('rowcnt', 'colcnt', 'loss_part', 'G16', 'gs_part'), ref, run(kind)):
      if name in ('loss_part', 'gs_part'):  # float sums: the kernels' instantiations may associate them differently
        assert (x - y).abs().max().item() <= 1e-5 * x.abs().max().item(), (kind, name)
      else:                                 # decisions and the bf16 operand: identical
        assert torch.equal(x, y), (kind, name)
