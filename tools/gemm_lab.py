"""NT-GEMM tile/epilogue lab on the config-B shapes: correctness of every tile variant against a torch fp32
reference, then interleaved timings (HIP events, random data).   python tools/gemm_lab.py [--rows 6976,3596]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16


def rnd(*shape, dtype=bf, scale=1.0):
  return (torch.randn(*shape, device=dev) * scale).to(dtype)


def gelu(x):
  return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def check(tile, M, N, K):
  R = ops.pad_rows(M)
  a, b = rnd(R, K), rnd(N, K, scale=0.05)
  bias, res = rnd(N, dtype=torch.float32), rnd(R, N, dtype=torch.float32)
  ref = a[:M].float() @ b.float().t()
  errs = []
  out = torch.zeros(R, N, device=dev, dtype=bf)
  out2 = torch.zeros(R, N, device=dev, dtype=bf)
  ops.gemm_nt(a, b, out, 'BIAS_GELU', m=M, bias=bias, out2=out2, tile=tile)
  errs.append((out[:M].float() - (ref + bias)).abs().max().item())
  errs.append((out2[:M].float() - gelu(out[:M].float())).abs().max().item())
  z = torch.zeros(R, N, device=dev)
  ops.gemm_nt(a, b, z, 'BIAS_DROP_RES', m=M, bias=bias, res=res, tile=tile)
  errs.append((z[:M] - (ref + bias + res[:M])).abs().max().item())
  aux = rnd(R, N)
  nblk = (M + 127) // 128
  cs = torch.zeros(nblk, N, device=dev)
  ops.gemm_nt(a, b, out, 'DGELU', m=M, aux=aux, colsum=cs, tile=tile)
  x = aux[:M].float().requires_grad_(True)
  gelu(x).sum().backward()
  want = ref * x.grad
  errs.append((out[:M].float() - want).abs().max().item() / max(1.0, want.abs().max().item()))
  errs.append((cs.sum(0) - out[:M].float().sum(0)).abs().max().item() / max(1.0, out[:M].float().sum(0).abs().max().item()))
  return errs


def timeit(fns, iters=20, rounds=3):
  """Each variant is captured as a HIP graph of `iters` back-to-back launches (the Python/ctypes launch path costs
  ~10 us per call, more than the small kernels); variants are replayed interleaved, best of `rounds`."""
  graphs = []
  side = torch.cuda.Stream()
  for f in fns:
    with torch.cuda.stream(side):
      for _ in range(2):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
      for _ in range(iters):
        f()
    graphs.append(g)
  torch.cuda.synchronize()
  best = [1e9] * len(fns)
  for _ in range(rounds):
    for i, g in enumerate(graphs):
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      g.replay()
      s.record()
      g.replay()
      e.record()
      torch.cuda.synchronize()
      best[i] = min(best[i], s.elapsed_time(e) / iters * 1e3)
  return best


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--rows', default='6976,3596')
  ap.add_argument('--tiles', default='1,2,3,4,5,6')
  ap.add_argument('--ablate', action='store_true', help='(needs a lab build with debug flags)')
  ap.add_argument('--nocheck', action='store_true')
  ap.add_argument('--instep', action='store_true',
                  help='launch as the training step does: M = the dense row count, live rows in a DEVICE scalar, row_index + device seed')
  ap.add_argument('--text', action='store_true', help='the text tower shapes (d = 768) instead of the video ones (d = 512)')
  args = ap.parse_args()
  tiles = [int(t) for t in args.tiles.split(',')]
  for tile in tiles:
    if tile < 3 or args.nocheck:
      continue
    for (M, N, K) in ([(300, 384, 128), (777, 576, 192), (1000, 768, 64)] if tile in (15, 16, 17, 20) else
                      [(300, 256, 128), (777, 512, 192), (1000, 768, 64), (6976, 3072, 512), (3639, 1536, 512), (5000, 1024, 1024)] if tile == 24 else
                      [(300, 256, 128), (777, 512, 192), (1000, 768, 64)]):
      errs = check(tile, M, N, K)
      ok = errs[0] < 0.1 and errs[1] < 0.05 and errs[2] < 2e-3 and errs[3] < 2e-2 and errs[4] < 2e-2
      print('check tile=%d %dx%dx%d errs=%s %s' % (tile, M, N, K, ' '.join('%.2e' % e for e in errs), 'OK' if ok else 'FAIL'))
  if args.ablate:
    for rows in [int(r) for r in args.rows.split(',')]:
      R = ops.pad_rows(rows)
      for (N, K, epi) in [(3072, 512, 'BIAS_BF16'), (512, 3072, 'ADD_F32')]:
        a, b = rnd(R, K), rnd(N, K, scale=0.05)
        bias, res = rnd(N, dtype=torch.float32), rnd(R, N, dtype=torch.float32)
        out = torch.zeros(R, N, device=dev, dtype=torch.float32 if epi == 'ADD_F32' else bf)
        out2 = torch.zeros(R, N, device=dev, dtype=bf)
        for tile in [t for t in tiles if t >= 3]:
          if tile in (4, 6) and N % 256:
            continue
          flags = [0, 16, 2, 18, 10, 26, 8, 24]
          fns = [lambda fl=fl: ops.gemm_nt(a, b, out, epi, m=rows, bias=bias, res=res, out2=out2, tile=tile | (fl << 8)) for fl in flags]
          ts = timeit(fns)
          print('ablate %5dx%4dx%4d %-10s tile=%d ' % (rows, N, K, epi, tile) +
                '  '.join('f%d %5.1f' % (fl, us) for fl, us in zip(flags, ts)))
    return
  DENSE = 6976
  for rows in [int(r) for r in args.rows.split(',')]:
    R = ops.pad_rows(DENSE if args.instep else rows)
    print('rows', rows, '(launched as the step does: M = %d, live rows on the device)' % DENSE if args.instep else '')
    nrd = torch.tensor([rows], device=dev, dtype=torch.int32)
    ridx = (torch.arange(R, device=dev, dtype=torch.int32) * 2) % (DENSE)
    seed = torch.tensor([7], device=dev, dtype=torch.int32)
    video = [(3072, 512, 'BIAS_GELU'), (3072, 512, 'DGELU'), (3072, 512, 'BIAS_BF16'), (1536, 512, 'BIAS_BF16'),
             (512, 512, 'BIAS_DROP_RES'), (512, 3072, 'BIAS_DROP_RES'), (512, 3072, 'ADD_F32'), (512, 1536, 'ADD_F32'),
             (512, 512, 'BF16')]
    text = [(3072, 768, 'BIAS_GELU'), (3072, 768, 'DGELU'), (2304, 768, 'BIAS_BF16'), (768, 768, 'BIAS_DROP_RES'),
            (768, 768, 'BF16')]
    for (N, K, epi) in (text if args.text else video):
      a, b = rnd(R, K), rnd(N, K, scale=0.05)
      bias, res = rnd(N, dtype=torch.float32), rnd(R, N, dtype=torch.float32)
      f32 = epi in ('BIAS_DROP_RES', 'ADD_F32')
      out = torch.zeros(R, N, device=dev, dtype=torch.float32 if f32 else bf)
      out2, aux = torch.zeros(R, N, device=dev, dtype=bf), rnd(R, N)
      cs = torch.zeros((rows + 127) // 128, N, device=dev)
      fns, used = [], []
      for tile in tiles:
        if (tile in (4, 6, 21) and N % 256) or (tile in (15, 16, 17, 20) and N % 192):
          continue
        kw = dict(bias=bias, res=res, out2=out2, aux=aux, tile=tile)
        if epi == 'BIAS_DROP_RES':
          kw.update(drop_key=1, drop_p=0.1)
        if epi == 'DGELU' and tile != 12:
          kw.update(colsum=cs)
        if args.instep:
          kw.update(n_rows_dev=nrd)
          if epi == 'BIAS_DROP_RES':
            kw.update(row_index=ridx, seed_dev=seed)
          fns.append(lambda kw=kw: ops.gemm_nt(a, b, out, epi, m=DENSE, **kw))
          used.append(tile)
          continue
        fns.append(lambda kw=kw: ops.gemm_nt(a, b, out, epi, m=rows, **kw))
        used.append(tile)
      ts = timeit(fns)
      fl = 2.0 * rows * N * K
      print('  %5dx%4dx%4d %-13s ' % (rows, N, K, epi) + '  '.join('t%d %6.1fus %4.0fTF' % (t, us, fl / us / 1e6) for t, us in zip(used, ts)))


if __name__ == '__main__':
  main()
