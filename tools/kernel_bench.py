"""Per-kernel timings on the config-B shapes (HIP events on the launch stream).  Usage on the GPU box:
   python tools/kernel_bench.py [--iters 20]
Prints one line per kernel: time, achieved TFLOP/s or GB/s."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402


def timeit(fn, iters):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / iters * 1e-3


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--iters', type=int, default=20)
  ap.add_argument('--rows', type=int, default=6976)
  args = ap.parse_args()
  dev = torch.device('cuda:0')
  rows = args.rows
  R = ops.pad_rows(rows)
  bf = torch.bfloat16

  def rnd(*shape, dtype=bf, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)

  print('rows', rows, 'padded', R)
  for (N, K, epi, tile) in [(1536, 512, 'BIAS_BF16', 0), (512, 512, 'BIAS_DROP_RES', 0), (3072, 512, 'BIAS_GELU', 0),
                            (512, 3072, 'BIAS_DROP_RES', 0), (3072, 512, 'DGELU', 0), (512, 1536, 'ADD_F32', 0),
                            (1536, 512, 'BIAS_BF16', 2), (3072, 512, 'BIAS_GELU', 2), (512, 3072, 'BIAS_DROP_RES', 1),
                            (512, 512, 'BIAS_DROP_RES', 1)]:
    a, b = rnd(R, K), rnd(N, K, scale=0.05)
    bias = rnd(N, dtype=torch.float32)
    res = rnd(R, N, dtype=torch.float32)
    f32 = epi in ('BIAS_DROP_RES', 'ADD_F32')
    out = torch.zeros(R, N, device=dev, dtype=torch.float32 if f32 else bf)
    out2 = torch.zeros(R, N, device=dev, dtype=bf)
    aux = rnd(R, N)
    kw = dict(bias=bias, res=res, out2=out2, aux=aux, tile=tile)
    if epi == 'BIAS_DROP_RES':
      kw.update(drop_key=1, drop_p=0.1)
    t = timeit(lambda: ops.gemm_nt(a, b, out, epi, m=rows, **kw), args.iters)
    print('gemm_nt %5dx%4dx%4d %-14s tile=%d  %8.1f us  %7.1f TF/s' % (rows, N, K, epi, tile, t * 1e6, 2.0 * rows * N * K / t / 1e12))
  for (N, K2, splits) in [(1536, 512, 4), (1536, 512, 8), (512, 512, 16), (3072, 512, 4), (3072, 512, 8), (512, 3072, 4),
                          (512, 3072, 8)]:
    a, b = rnd(R, N), rnd(R, K2)
    out = torch.zeros(N, K2, device=dev)
    t = timeit(lambda: ops.gemm_tn(a, b, rows=rows, splits=splits, out=out), args.iters)
    print('gemm_tn %5d rows -> %4dx%4d splits=%2d  %8.1f us  %7.1f TF/s' % (rows, N, K2, splits, t * 1e6, 2.0 * rows * N * K2 / t / 1e12))
  B, S, H = 32, 218, 4
  d = H * 128
  qkv = rnd(R, 3 * d)
  bias = torch.zeros(R, device=dev)
  scale = 1.0 / math.sqrt(128.0)
  for p in (0.0, 0.1):
    ctx, lse = ops.attn_fwd(qkv, bias, B, S, H, scale, drop_key=3, drop_p=p)
    t = timeit(lambda: ops.attn_fwd(qkv, bias, B, S, H, scale, drop_key=3, drop_p=p), args.iters)
    fl = 4.0 * B * H * S * S * 128
    print('attn_fwd p=%.1f  %8.1f us  %7.1f TF/s' % (p, t * 1e6, fl / t / 1e12))
    dctx = rnd(R, d)
    t = timeit(lambda: ops.attn_bwd(qkv, bias, ctx, lse, dctx, B, S, H, scale, drop_key=3, drop_p=p), args.iters)
    print('attn_bwd p=%.1f  %8.1f us  %7.1f TF/s (2.5x fwd flops)' % (p, t * 1e6, 2.5 * fl / t / 1e12))
  z = rnd(R, d, dtype=torch.float32)
  g, be = torch.ones(d, device=dev), torch.zeros(d, device=dev)
  t = timeit(lambda: ops.ln_fwd(z, g, be, 1e-12, rows=rows), args.iters)
  print('ln_fwd (incl. torch allocs)  %8.1f us' % (t * 1e6))


if __name__ == '__main__':
  main()
