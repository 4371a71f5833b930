"""Thin per-kernel Python wrappers over the C ABI (used by the unit tests and by the modules).

Every function takes CUDA(HIP) torch tensors, passes raw device pointers + the current stream to
libmmt_hip.so and returns torch tensors.  No computation happens in Python and nothing here falls
back to torch ops.
"""
import ctypes

import torch

from . import _lib
from ._lib import EPI, MmtEpilogue, check

ROW_ALIGN = 256


def pad_rows(n):
  return (n + ROW_ALIGN - 1) // ROW_ALIGN * ROW_ALIGN


def _p(t):
  return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
  for t in ts:
    if t is not None and not t.is_cuda:
      raise RuntimeError('mmt_amd kernels need GPU tensors (no CPU fallback)')


def dropout_params(p):
  """(thr16, scale) with keep-probability quantised to 1/65536 (see mmt_common.h)."""
  thr = int(round(float(p) * 65536.0))
  if thr <= 0:
    return 0, 1.0
  return thr, 1.0 / (1.0 - thr / 65536.0)


def gemm_nt(a, b, out, epilogue='BF16', m=None, bias=None, res=None, out2=None, aux=None, colsum=None,
            row_index=None, drop_key=0, drop_p=0.0, n_rows_dev=None, tile=0, seed_dev=None, live_rows=0):
  """out[M,N] = a[M,K] @ b[N,K]^T with a fused epilogue.  a/b bf16 row-major, rows padded to 128.  live_rows: the host's
  count of the rows n_rows_dev will report (tile choice only, MmtEpilogue.live_rows_hint; 0 = unknown)."""
  _need_cuda(a, b, out)
  M = a.shape[0] if m is None else m
  N, K = b.shape
  e = MmtEpilogue()
  e.bias, e.res, e.out2, e.aux = (x.data_ptr() if x is not None else None for x in (bias, res, out2, aux))
  e.ldres = res.stride(0) if res is not None else 0
  e.ldout2 = out2.stride(0) if out2 is not None else 0
  e.ldaux = aux.stride(0) if aux is not None else 0
  e.colsum = colsum.data_ptr() if colsum is not None else None
  e.row_index = row_index.data_ptr() if row_index is not None else None
  thr, scale = dropout_params(drop_p)
  e.drop_key, e.drop_thr16, e.drop_scale = drop_key, thr, scale
  e.reserved = tile
  e.live_rows_hint = int(live_rows or 0)
  e.seed_dev = seed_dev.data_ptr() if seed_dev is not None else None
  rc = _lib.lib().mmt_gemm_nt_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K,
                                   EPI[epilogue], ctypes.byref(e), _p(n_rows_dev), _stream())
  check(rc, 'mmt_gemm_nt_bf16')
  return out


def gemm_nn(a, w, out, epilogue='BF16', m=None, res=None, aux=None, n_rows_dev=None, tile=0):
  """out[M,N] = a[M,K] @ w[K,N] (w row-major [K, N]: a weight [out, in] as stored -> input gradient), fused epilogue
  BF16 / F32 / ADD_F32 / DGELU."""
  _need_cuda(a, w, out)
  M = a.shape[0] if m is None else m
  K, N = w.shape
  e = MmtEpilogue()
  e.res = res.data_ptr() if res is not None else None
  e.ldres = res.stride(0) if res is not None else 0
  e.aux = aux.data_ptr() if aux is not None else None
  e.ldaux = aux.stride(0) if aux is not None else 0
  e.reserved = tile
  check(_lib.lib().mmt_gemm_nn_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), M, N, K,
                                    EPI[epilogue], ctypes.byref(e), _p(n_rows_dev), _stream()), 'mmt_gemm_nn_bf16')
  return out


def gemm_tn(a, b, rows=None, splits=4, out=None, accumulate=False, n_rows_dev=None):
  """out[N,K2] (+)= a[rows,N]^T @ b[rows,K2]  (fp32 result; split over rows, deterministic reduce)."""
  _need_cuda(a, b)
  rows = a.shape[0] if rows is None else rows
  N, K2 = a.shape[1], b.shape[1]
  ws = torch.empty(splits, N, K2, device=a.device, dtype=torch.float32)
  L = _lib.lib()
  check(L.mmt_gemm_tn_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(ws), rows, N, K2, splits,
                           _p(n_rows_dev), _stream()), 'mmt_gemm_tn_bf16')
  if out is None:
    out = torch.empty(N, K2, device=a.device, dtype=torch.float32)
    accumulate = False
  check(L.mmt_reduce_slabs(_p(ws), splits, N * K2, _p(out), int(accumulate), _stream()), 'mmt_reduce_slabs')
  return out


def gemm_nt_splitk(a, b, out, epilogue='F32', m=None, bias=None, res=None, row_index=None, drop_key=0, drop_p=0.0,
                   seed_dev=None, splits=0, wide=False, n_rows_dev=None, no_epilogue=False, ws=None, b_kn=False):
  """Split-K variant of gemm_nt for skinny problems (few rows, long K).  b_kn: b is [K, N] row-major (gemm_nn)."""
  _need_cuda(a, b, out)
  M = a.shape[0] if m is None else m
  N, K = (b.shape[1], b.shape[0]) if b_kn else b.shape
  e = MmtEpilogue()
  e.bias = bias.data_ptr() if bias is not None else None
  e.res = res.data_ptr() if res is not None else None
  e.ldres = res.stride(0) if res is not None else 0
  e.row_index = row_index.data_ptr() if row_index is not None else None
  thr, scale = dropout_params(drop_p)
  e.drop_key, e.drop_thr16, e.drop_scale = drop_key, thr, scale
  e.seed_dev = seed_dev.data_ptr() if seed_dev is not None else None
  L = _lib.lib()
  if ws is None:
    ws = torch.empty(L.mmt_gemm_nt_splitk_workspace_floats(M, N, K), device=a.device, dtype=torch.float32)
  fn = L.mmt_gemm_nn_splitk_ex if b_kn else L.mmt_gemm_nt_splitk_ex
  check(fn(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, EPI[epilogue],
                                ctypes.byref(e), _p(ws), int(splits), int(wide),
                                _p(n_rows_dev) if n_rows_dev is not None else None, int(no_epilogue), _stream()),
        'mmt_gemm_nt_splitk_ex')
  return out


def gemm_nt_grouped(items, m=None, epilogue='BIAS_F32', n_rows_dev=None, n_rows_index=None):
  """items: list of (a [M,K] bf16, b [N,K] bf16, out [M,N] fp32, bias [N] fp32 or None): all in ONE launch.
  n_rows_dev: int32 device tensor of live rows per problem (tiles past them exit), or None; item i reads entry
  n_rows_index[i] (default i)."""
  from ._lib import MmtGemmItem
  arr = (MmtGemmItem * len(items))()
  for i, (a, b, out, bias) in enumerate(items):
    _need_cuda(a, b, out)
    it = arr[i]
    j = i if n_rows_index is None else n_rows_index[i]
    it.n_rows_dev = (n_rows_dev.data_ptr() + 4 * j) if n_rows_dev is not None else None
    it.A, it.B, it.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    it.bias = bias.data_ptr() if bias is not None else None
    it.lda, it.ldb, it.ldc = a.stride(0), b.stride(0), out.stride(0)
    it.M, it.N, it.K = (a.shape[0] if m is None else m), b.shape[0], b.shape[1]
  check(_lib.lib().mmt_gemm_nt_grouped(arr, len(items), EPI[epilogue], _stream()), 'mmt_gemm_nt_grouped')


def wgrad_grouped(items, rows, n_rows_dev=None, item_rows_dev=None):
  """items: list of (a [rows,N] bf16, b [rows,K2] bf16, out fp32 [N_out,K2_out], bias_out fp32 [N_out] or None).
  One launch: out = a^T @ b (fp32), bias_out = column sums of a.  n_rows_dev: one live row count for all items;
  item_rows_dev: int32 device tensor [len(items)], one per item."""
  from ._lib import MmtWgradGroup
  g = MmtWgradGroup()
  g.count, g.rows = len(items), rows
  g.n_rows_dev = n_rows_dev.data_ptr() if n_rows_dev is not None else None
  for i, (a, b, out, bias) in enumerate(items):
    _need_cuda(a, b, out)
    it = g.item[i]
    it.n_rows_dev = (item_rows_dev.data_ptr() + 4 * i) if item_rows_dev is not None else None
    it.A, it.B, it.out = a.data_ptr(), b.data_ptr(), out.data_ptr()
    it.bias_out = bias.data_ptr() if bias is not None else None
    it.lda, it.ldb, it.ldo = a.stride(0), b.stride(0), out.stride(0)
    it.N, it.K2, it.N_out, it.K2_out = a.shape[1], b.shape[1], out.shape[0], out.shape[1]
  check(_lib.lib().mmt_wgrad_grouped(ctypes.byref(g), _stream()), 'mmt_wgrad_grouped')


def sgemm_batched(As, Bs, Cs, M, N, K, sai, sak, sbj, sbk, ldc, biases=None, beta=0.0):
  """C_b[i][j] = beta*C_b[i][j] + sum_k A_b[i*sai + k*sak] * B_b[j*sbj + k*sbk] (+ bias_b[j]); fp32, exact-fp32 MFMA."""
  from ._lib import MmtSgemm
  _need_cuda(*As, *Bs, *Cs)
  g = MmtSgemm()
  g.batch, g.M, g.N, g.K = len(As), M, N, K
  g.sai, g.sak, g.sbj, g.sbk, g.ldc, g.beta = sai, sak, sbj, sbk, ldc, beta
  for i in range(len(As)):
    g.A[i], g.B[i], g.C[i] = As[i].data_ptr(), Bs[i].data_ptr(), Cs[i].data_ptr()
    g.bias[i] = biases[i].data_ptr() if biases is not None and biases[i] is not None else None
  check(_lib.lib().mmt_sgemm_batched(ctypes.byref(g), _stream()), 'mmt_sgemm_batched')
  return Cs


def ln_fwd(z, gamma, beta, eps, rows=None, n_rows_dev=None, want_h32=True):
  _need_cuda(z)
  R, d = z.shape
  rows = R if rows is None else rows
  h32 = torch.zeros_like(z) if want_h32 else None
  h16 = torch.zeros(R, d, device=z.device, dtype=torch.bfloat16)
  mean = torch.zeros(R, device=z.device, dtype=torch.float32)
  rstd = torch.zeros(R, device=z.device, dtype=torch.float32)
  check(_lib.lib().mmt_ln_fwd(_p(z), _p(gamma), _p(beta), eps, _p(h32), _p(h16), _p(mean), _p(rstd), rows, d,
                              _p(n_rows_dev), _stream()), 'mmt_ln_fwd')
  return h32, h16, mean, rstd


def gemm_nt_ln_fwd(a, w, bias, res, gamma, beta, eps, m=None, row_index=None, drop_key=0, drop_p=0.0, seed_dev=None,
                   n_rows_dev=None):
  """z = dropout(a @ w^T + bias) + res ; h = LN(z) in ONE launch (model/bert.py:185-188).  -> z, h32, h16, mean, rstd"""
  _need_cuda(a, w, res)
  M = a.shape[0] if m is None else m
  N, K = w.shape
  R = res.shape[0]
  z = torch.zeros(R, N, device=a.device, dtype=torch.float32)
  h32 = torch.zeros(R, N, device=a.device, dtype=torch.float32)
  h16 = torch.zeros(R, N, device=a.device, dtype=torch.bfloat16)
  mean = torch.zeros(R, device=a.device, dtype=torch.float32)
  rstd = torch.zeros(R, device=a.device, dtype=torch.float32)
  thr, scale = dropout_params(drop_p)
  check(_lib.lib().mmt_gemm_nt_ln_fwd(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(res), res.stride(0), _p(row_index),
                                      drop_key, thr, scale, _p(seed_dev), _p(z), _p(gamma), _p(beta), eps, _p(h32), _p(h16),
                                      _p(mean), _p(rstd), M, N, K, _p(n_rows_dev), _stream()), 'mmt_gemm_nt_ln_fwd')
  return z, h32, h16, mean, rstd


def embed_ln_fwd(features, type_ids, pos_ids, type_emb, pos_emb, gamma, beta, eps, rows=None, drop_key=0,
                 drop_p=0.0, row_index=None, n_rows_dev=None, seed_dev=None):
  _need_cuda(features)
  R, d = features.shape
  rows = R if rows is None else rows
  z = torch.zeros_like(features)
  h32 = torch.zeros_like(features)
  h16 = torch.zeros(R, d, device=features.device, dtype=torch.bfloat16)
  mean = torch.zeros(R, device=features.device, dtype=torch.float32)
  rstd = torch.zeros(R, device=features.device, dtype=torch.float32)
  thr, scale = dropout_params(drop_p)
  check(_lib.lib().mmt_embed_ln_fwd(_p(features), _p(type_ids), _p(pos_ids), _p(type_emb), _p(pos_emb), _p(z),
                                    _p(gamma), _p(beta), eps, _p(h32), _p(h16), _p(mean), _p(rstd), rows, d,
                                    _p(n_rows_dev), _p(row_index), drop_key, thr, scale, _p(seed_dev), _stream()),
        'mmt_embed_ln_fwd')
  return z, h32, h16, mean, rstd


def ln_bwd(dout, z, mean, rstd, gamma, rows=None, drop_mode=0, drop_key=0, drop_p=0.0, want_dy=True,
           row_index=None, n_rows_dev=None, seed_dev=None):
  """Returns dz (fp32), dy (bf16 or None), dgamma, dbeta, dbias (fp32 [d])."""
  _need_cuda(dout)
  R, d = z.shape
  rows = R if rows is None else rows
  L = _lib.lib()
  rpb = L.mmt_ln_bwd_rows_per_block(rows)
  nblk = (rows + rpb - 1) // rpb
  dz = torch.zeros_like(z)
  dy = torch.zeros(R, d, device=z.device, dtype=torch.bfloat16) if want_dy else None
  partials = torch.empty(nblk, 3, d, device=z.device, dtype=torch.float32)
  thr, scale = dropout_params(drop_p)
  check(L.mmt_ln_bwd(_p(dout), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dz), _p(dy), _p(partials), rows, d,
                     drop_mode, _p(n_rows_dev), _p(row_index), drop_key, thr, scale, _p(seed_dev), _stream()), 'mmt_ln_bwd')
  outs = [torch.empty(d, device=z.device, dtype=torch.float32) for _ in range(3)]
  check(L.mmt_col_reduce(_p(partials), nblk, 3, d, _p(outs[0]), _p(outs[1]), _p(outs[2]), None, 0, _stream()),
        'mmt_col_reduce')
  return dz, dy, outs[0], outs[1], outs[2]


def table_grad(g, ids, vocab, rows=None, n_rows_dev=None):
  _need_cuda(g)
  R, d = g.shape
  rows = R if rows is None else rows
  out = torch.empty(vocab, d, device=g.device, dtype=torch.float32)
  L = _lib.lib()
  scratch = torch.empty(L.mmt_table_grad_scratch_floats(vocab, d), device=g.device, dtype=torch.float32)
  check(L.mmt_table_grad(_p(g), _p(ids), rows, d, vocab, _p(n_rows_dev), _p(scratch), _p(out), 0, _stream()),
        'mmt_table_grad')
  return out


def attn_fwd(qkv, mask_bias, B, S, H, scale, cu_seqlens=None, drop_key=0, drop_p=0.0, seed_dev=None, row_index=None):
  _need_cuda(qkv)
  R, d3 = qkv.shape
  d = d3 // 3
  ctx = torch.zeros(R, d, device=qkv.device, dtype=torch.bfloat16)
  lse = torch.zeros(R, H, device=qkv.device, dtype=torch.float32)
  thr, sc = dropout_params(drop_p)
  check(_lib.lib().mmt_attn_fwd(_p(qkv), _p(cu_seqlens), _p(mask_bias), _p(ctx), _p(lse), B, S, H, d, scale,
                                drop_key, thr, sc, _p(seed_dev), _p(row_index), _stream()), 'mmt_attn_fwd')
  return ctx, lse


def attn_bwd(qkv, mask_bias, ctx, lse, dctx, B, S, H, scale, cu_seqlens=None, drop_key=0, drop_p=0.0,
             seed_dev=None, row_index=None, scheduled=False):
  """scheduled: run the blocks in the order of mmt_attn_schedule (packed batches with (B * H) % 8 == 0): same results."""
  _need_cuda(qkv)
  R, d3 = qkv.shape
  d = d3 // 3
  dqkv = torch.zeros_like(qkv)
  delta = torch.zeros(R, d // 64, device=qkv.device, dtype=torch.float32)  # scratch: dO * O sums per 64 columns
  thr, sc = dropout_params(drop_p)
  L = _lib.lib()
  if scheduled:
    work = torch.empty(L.mmt_attn_schedule_words(B, S, H), device=qkv.device, dtype=torch.int32)
    check(L.mmt_attn_schedule(_p(cu_seqlens), B, S, H, _p(work), _stream()), 'mmt_attn_schedule')
    check(L.mmt_attn_bwd_ex(_p(qkv), _p(cu_seqlens), _p(mask_bias), _p(ctx), _p(lse), _p(dctx), _p(dqkv), _p(delta), 0, B, S, H, d,
                            scale, drop_key, thr, sc, _p(seed_dev), _p(row_index), _p(work), _stream()), 'mmt_attn_bwd_ex')
    return dqkv
  check(L.mmt_attn_bwd(_p(qkv), _p(cu_seqlens), _p(mask_bias), _p(ctx), _p(lse), _p(dctx), _p(dqkv),
                       _p(delta), B, S, H, d, scale, drop_key, thr, sc, _p(seed_dev), _p(row_index), _stream()), 'mmt_attn_bwd')
  return dqkv


def attn_dropout_mask(B, H, S, drop_key, drop_p, device='cuda', seed_dev=None):
  out = torch.empty(B, H, S, S, device=device, dtype=torch.uint8)
  thr, _ = dropout_params(drop_p)
  check(_lib.lib().mmt_attn_dropout_mask(_p(out), B, H, S, drop_key, thr, _p(seed_dev), _stream()), 'mmt_attn_dropout_mask')
  return out
