"""Row-sharded global similarity + max-margin ranking loss for very large global batches (BASELINE.json configs[4]:
64k pairs over 8 ranks).

`GraphedTrainStep` makes every rank build the whole n x n similarity, which is the right trade for n of a few hundred
pairs (microseconds) and impossible at n = 65 536 (17 GB of fp32 plus the reference's [n, n, M] weights).  Here rank r
owns the text ROWS r0..r0+b: its b texts against all n (all-gathered) videos,

    S = (tw (.) T) (vw (.) V)^T / den      one bf16 MFMA GEMM with K = M*d          (model/model.py:789-837)
    loss, dL/dS                            two passes over the row block + an n-vector all-reduce (model/loss.py:38-65)
    dT, dtw   from P = G' V'               one GEMM
    dV        from Q = G'^T T'             one GEMM + a reduce-scatter over ranks

The maths is phase-structured (`phase_*`) so that the collectives sit between plain function calls; `ShardedSimLoss`
wires the phases to torch.distributed (RCCL) and degenerates to a single row block without a process group.
"""
import os

import torch
import torch.distributed as dist

from . import _lib, ops
from ._lib import check


def _f32(x):
  return x.detach().contiguous().float()


def _transposed(L, x):
  """x [r, c] bf16 (r, c multiples of 128) -> x^T [c, r], contiguous."""
  r, c = x.shape
  out = torch.empty(c, r, device=x.device, dtype=torch.bfloat16)
  check(L.mmt_transpose_bf16(ops._p(x), c, r, c, ops._p(out), r, ops._stream()), 'mmt_transpose_bf16')
  return out


class RowBlock:
  """State of one rank's row block between phases."""

  def __init__(self, txt, tw, vid_all, vw_all, r0, margin, fix_norm=True, keep_similarity=False):
    """txt [b, M, d], tw [b, M]: local texts; vid_all [n, M, d], vw_all [n, M]: every rank's videos (rank order).
    keep_similarity: `self.S` holds the similarities after phase_counts (an extra write of the row block); by default it keeps
    the raw numerators of the GEMM and both passes divide on the fly (`similarity()` returns a finished copy)."""
    self.keep_similarity = bool(keep_similarity)
    self.txt, self.tw, self.vid_all, self.vw_all = _f32(txt), _f32(tw), _f32(vid_all), _f32(vw_all)
    self.b, self.m, self.d = self.txt.shape
    self.n = self.vid_all.shape[0]
    self.r0, self.margin, self.fix_norm = int(r0), float(margin), bool(fix_norm)
    if self.n % 128 or (self.m * self.d) % 128:
      raise ValueError('row-sharded similarity needs n % 128 == 0 and (M*d) % 128 == 0')
    self.norm = 2.0 * self.n * (self.n - 1) if fix_norm else 2.0 * self.n * self.n
    self.dev = self.txt.device
    self.L = _lib.lib()
    self.vw_t = self.vw_all.t().contiguous()  # [M, n]: the layout the two sweeps read (coalesced per expert)

  # ---- phase A: similarity row block -------------------------------------------------------------
  def phase_similarity(self):
    L, b, n, m, d = self.L, self.b, self.n, self.m, self.d
    md, bp = m * d, ops.pad_rows(b)
    self.t16 = torch.empty(bp, md, device=self.dev, dtype=torch.bfloat16)
    self.v16 = torch.empty(n, md, device=self.dev, dtype=torch.bfloat16)
    check(L.mmt_ls_fold_bf16(ops._p(self.txt), ops._p(self.tw), b, bp, m, d, ops._p(self.t16), ops._stream()), 'mmt_ls_fold_bf16')
    check(L.mmt_ls_fold_bf16(ops._p(self.vid_all), ops._p(self.vw_all), n, n, m, d, ops._p(self.v16), ops._stream()),
          'mmt_ls_fold_bf16')
    pad = int(os.environ.get('MMT_LS_PAD', '0'))  # lab: leading dimension of the row block = n + pad
    self.S = torch.empty(bp, n + pad, device=self.dev, dtype=torch.float32)[:, :n]
    self.ld = self.S.stride(0)
    ops.gemm_nt(self.t16, self.v16, self.S, 'F32', m=b)
    # S holds the raw numerators until phase_counts divides them in its own sweep; the diagonal needs b divisions now
    self.diag_local = torch.empty(b, device=self.dev, dtype=torch.float32)
    check(L.mmt_ls_diag(ops._p(self.S), self.ld, ops._p(self.tw), ops._p(self.vw_all), b, n, m, self.r0, ops._p(self.diag_local),
                        ops._stream()), 'mmt_ls_diag')
    return self.diag_local

  # ---- phase B: hinge counts (needs the global diagonal) -------------------------------------------
  def phase_counts(self, diag_all):
    L, b, n = self.L, self.b, self.n
    self.diag_all = _f32(diag_all)
    self.rowcnt = torch.zeros(b, device=self.dev, dtype=torch.int32)
    self.colcnt = torch.zeros(n, device=self.dev, dtype=torch.int32)
    part = torch.empty(b, L.mmt_ls_col_blocks(n), device=self.dev, dtype=torch.float32)
    # numerators -> similarities and pass 1 (hinge sums and counts) in ONE sweep over the row block
    check(L.mmt_ls_counts_ex(ops._p(self.S), self.ld, ops._p(self.diag_all), ops._p(self.tw), ops._p(self.vw_all), ops._p(self.vw_t),
                             self.m, 1 if self.keep_similarity else 2, b, n,
                             self.r0, self.margin, ops._p(self.rowcnt), ops._p(self.colcnt), ops._p(part), ops._stream()),
          'mmt_ls_counts_ex')
    self.loss_part = part.sum(1)  # per row, column blocks in order
    return self.colcnt, self.loss_part.sum() / self.norm

  def similarity(self):
    """The [b, n] similarities of the block (after phase_similarity)."""
    if self.keep_similarity and hasattr(self, 'rowcnt'):
      return self.S[:self.b]
    s = self.S[:self.b].contiguous().clone() if self.ld != self.n else self.S[:self.b].clone()
    check(self.L.mmt_ls_finish(ops._p(s), self.n, ops._p(self.tw), ops._p(self.vw_all), self.b, self.n, self.m, ops._stream()),
          'mmt_ls_finish')
    return s

  # ---- phase C: gradients of the local texts, contribution to every video ----------------------------
  def phase_backward(self, colcnt_total):
    L, b, n, m, d = self.L, self.b, self.n, self.m, self.d
    md, bp = m * d, self.t16.shape[0]
    colcnt_total = colcnt_total.to(device=self.dev, dtype=torch.int32).contiguous()
    g16 = torch.empty(bp, n, device=self.dev, dtype=torch.bfloat16)
    g16[b:].zero_()                                        # (pad rows: K of the Q product)
    gs_part = torch.empty(b, L.mmt_ls_col_blocks(n), m, device=self.dev, dtype=torch.float32)
    check(L.mmt_ls_grad_ex(ops._p(self.S), self.ld, ops._p(self.diag_all), ops._p(self.tw), ops._p(self.vw_all), ops._p(self.vw_t),
                           ops._p(self.rowcnt), ops._p(colcnt_total), b, n, m, self.r0, self.margin, 1.0 / self.norm, ops._p(g16), n, ops._p(gs_part),
                           0 if self.keep_similarity else 1, ops._stream()), 'mmt_ls_grad_ex')
    gs = gs_part.sum(1)
    v16t = _transposed(L, self.v16)                        # [M*d, n]: B operand of P = G' V'
    p = torch.empty(bp, md, device=self.dev, dtype=torch.float32)
    ops.gemm_nt(g16, v16t, p, 'F32', m=b)
    dtxt = torch.empty_like(self.txt)
    dtw = torch.empty_like(self.tw)
    check(L.mmt_ls_unfold(ops._p(p), md, ops._p(self.txt), ops._p(self.tw), ops._p(gs), b, m, d, ops._p(dtxt), ops._p(dtw),
                          ops._stream()), 'mmt_ls_unfold')
    q = torch.empty(n, md, device=self.dev, dtype=torch.float32)   # Q = G'^T T': this rank's share of every video's gradient
    ops.gemm_nt(_transposed(L, g16), _transposed(L, self.t16), q, 'F32')   # K = the (zero-padded) local rows
    return dtxt, dtw, q

  # ---- phase D: gradient of the local videos from the reduce-scattered Q rows ---------------------------
  def phase_video_grad(self, q_rows, vid_local, vw_local):
    vid_local, vw_local = _f32(vid_local), _f32(vw_local)
    b, m, d = vid_local.shape
    dvid = torch.empty_like(vid_local)
    q_rows = q_rows.contiguous()
    check(self.L.mmt_ls_unfold(ops._p(q_rows), m * d, ops._p(vid_local), ops._p(vw_local), None, b, m, d, ops._p(dvid), None,
                               ops._stream()), 'mmt_ls_unfold')
    return dvid


class _ShardedSimLossFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, vid, txt, vw, tw, margin, fix_norm, group):
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    b = vid.shape[0]

    def gather(x):
      if world == 1:
        return x.detach().contiguous()
      out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
      dist.all_gather_into_tensor(out, x.detach().contiguous(), group=group)
      return out

    blk = RowBlock(txt, tw, gather(vid), gather(vw), rank * b, margin, fix_norm)
    diag_all = gather(blk.phase_similarity())
    colcnt, loss = blk.phase_counts(diag_all)
    if world > 1:
      dist.all_reduce(colcnt, group=group)
      dist.all_reduce(loss, group=group)
    if not fix_norm:
      loss = loss + 2.0 * blk.n * max(float(margin), 0.0) / blk.norm
    dtxt, dtw, q = blk.phase_backward(colcnt)
    if world > 1:
      q_rows = torch.empty(b, q.shape[1], device=q.device, dtype=q.dtype)
      dist.reduce_scatter_tensor(q_rows, q, group=group)
    else:
      q_rows = q
    dvid = blk.phase_video_grad(q_rows, vid, vw)
    ctx.save_for_backward(dvid, dtxt, dtw)
    return loss

  @staticmethod
  def backward(ctx, gout):
    dvid, dtxt, dtw = ctx.saved_tensors
    return dvid * gout, dtxt * gout, None, dtw * gout, None, None, None


class ShardedSimLoss(torch.nn.Module):
  """loss = MaxMarginRankingLoss(margin, fix_norm)(sharded_cross_view_inner_product(...)) over the GLOBAL batch,
  with the n x n matrix sharded by text rows over the ranks of `group`.  Inputs are this rank's (b, M, d) expert
  embeddings (one caption per video, as in training) and (b, M) mixture weights; video weights carry no gradient
  (vid_wgh='none' in every published config)."""

  def __init__(self, margin=0.05, fix_norm=True, group=None):
    super().__init__()
    self.margin, self.fix_norm, self.group = margin, fix_norm, group

  def forward(self, vid_embds, text_embds, vid_weights, text_weights):
    if text_embds.dim() == 4:  # (B, M, C=1, d) as CENet returns it
      if text_embds.shape[2] != 1:
        raise NotImplementedError('row-sharded loss: one caption per video (training layout)')
      text_embds = text_embds[:, :, 0]
    if text_weights.dim() == 3:
      text_weights = text_weights[:, 0]
    if not vid_embds.is_cuda:
      raise RuntimeError('mmt_amd.large_sim runs on the GPU only (no CPU fallback)')
    if vid_weights.requires_grad:
      raise NotImplementedError('row-sharded loss: video mixture weights are constants (vid_wgh="none")')
    return _ShardedSimLossFn.apply(vid_embds, text_embds, vid_weights, text_weights, self.margin, self.fix_norm, self.group)
