"""Attention kernel lab: forward / backward time as a function of sequence length and batch (HIP-graph replay timing),
to separate per-block latency from throughput.  python tools/attn_lab.py

Findings so far (B = 32, S = 218, head dim 128: 17.8 us forward with the shipped kernel):
  * a 4-deep prefetch ring (all key tiles requested up front, 128 KiB LDS, 1 block/CU): 27 us -- co-resident blocks
    matter more than prefetch depth;
  * split over keys (16 queries x 4 waves, one key tile per wave, K fragments straight from global, wave-private V tile,
    log-sum-exp merge): 38 us -- every 16-query group re-stages K/V;
  * transpose reads from inline asm (so the compiler's wait-count pass no longer drains the next tile's prefetch with a
    vmcnt(0) in front of them) + the key-mask bias fetched once in the prologue: 18.7 us -- the exposed DMA latency was
    not the limiter either.
  * 8 waves per block, the wave pair (w, w+4) sharing 16 queries and splitting every 64-key tile in halves (same staged
    tile, half the MFMA / softmax work per wave, merge at the end): 19.5 us -- halving the per-wave instruction stream
    changes nothing, so the limiter is a per-CU resource.  PMC (profiles/r01_pmc_kernels.csv, first pass): 52 % of the
    wave cycles are parked in s_waitcnt / barriers, LDS bank-conflict cycles are 28 % of the LDS-active cycles.
  Throughput view (B = 512: 206 us = 3800 CU-cycles per 64x64x128 tile step, MFMA needs 512): each wave runs its
  MFMAs (512 cycles), the softmax / dropout VALU work (~1200), the LDS-DMA issue (~800) and LDS waits one after the
  other, with only two waves per SIMD to overlap them; the VALU share is the next thing to cut."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import _lib, ops  # noqa: E402
from mmt_amd.ops import _p, _stream  # noqa: E402
from tools.gemm_lab import timeit  # noqa: E402

dev = torch.device('cuda:0')
H, d = 4, 512


def case(B, S, drop_p):
  rows = B * S
  R = ops.pad_rows(rows)
  qkv = (torch.randn(R, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
  mask = torch.zeros(R, device=dev)
  cu = torch.arange(0, B + 1, device=dev, dtype=torch.int32) * S
  ctx = torch.zeros(R, d, device=dev, dtype=torch.bfloat16)
  lse = torch.zeros(R, H, device=dev)
  dctx = (torch.randn(R, d, device=dev) * 0.1).to(torch.bfloat16)
  dqkv = torch.zeros_like(qkv)
  delta = torch.zeros(R, d // 64, device=dev)
  thr, sc = ops.dropout_params(drop_p)
  L = _lib.lib()
  scale = 128 ** -0.5

  def fwd():
    _lib.check(L.mmt_attn_fwd(_p(qkv), _p(cu), _p(mask), _p(ctx), _p(lse), B, S, H, d, scale, 7, thr, sc, None, None, _stream()), 'f')

  def bwd():
    _lib.check(L.mmt_attn_bwd(_p(qkv), _p(cu), _p(mask), _p(ctx), _p(lse), _p(dctx), _p(dqkv), _p(delta), B, S, H, d, scale,
                              7, thr, sc, None, None, _stream()), 'b')

  torch.cuda.synchronize()  # inputs were produced on the default stream; timeit launches on a side stream
  tf, tb = timeit([fwd, bwd])
  fl = 4.0 * B * H * S * S * 128
  print('B %4d S %4d p %.1f | fwd %6.1f us (%5.1f TF)  bwd(dq+dkv) %6.1f us (%5.1f TF)' %
        (B, S, drop_p, tf, fl / tf / 1e6, tb, 2.5 * fl / tb / 1e6))


for S in (32, 64, 128, 218):
  for B in (32, 128, 512):
    case(B, S, 0.1)
case(32, 218, 0.0)
case(128, 128, 0.0)
