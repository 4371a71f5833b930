"""Cycle budget of the persistent wave-specialised GEMM (gemm5.hip, tile 24), per role.  Needs the instrumented lab build:
   python -m mmt_amd.build --instr
   MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so python tools/g5_budget.py [rows]
Per block, one wave of every role reports s_memtime sums: consumer groups -- K-loop work (fragment reads + MFMAs) /
K-loop barrier waits / epilogue work (staging + sweep + stores) / epilogue barrier waits; producers -- prologue, LDS-DMA
issue, vmcnt waits, barrier waits.  Launched as the step launches it (dense-sized M, live row count on the device)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16


def rnd(*shape, dtype=bf, scale=1.0):
  return (torch.randn(*shape, device=dev) * scale).to(dtype)


def budget(name, live, dense, N, K, epi, tile=24, cold=False, drop=0.0):
  R = ops.pad_rows(dense)
  a, b = rnd(R, K), rnd(N, K, scale=0.05)
  out = torch.zeros(R, N, device=dev, dtype=torch.float32 if epi in ('BIAS_DROP_RES', 'ADD_F32', 'F32') else bf)
  nrd = torch.tensor([live], device=dev, dtype=torch.int32)
  extra = {}
  if epi in ('BIAS_GELU', 'BIAS_DROP_RES', 'BIAS_BF16'):
    extra['bias'] = rnd(N, dtype=torch.float32)
  if epi == 'BIAS_GELU':
    extra['out2'] = torch.zeros(R, N, device=dev, dtype=bf)
  if epi in ('BIAS_DROP_RES', 'ADD_F32'):
    extra['res'] = rnd(R, N, dtype=torch.float32)
  if epi == 'DGELU':
    extra['aux'] = rnd(R, N)
  if drop:
    extra.update(drop_key=1234, drop_p=drop)
  dbg = torch.zeros(256 * 3, 8, device=dev, dtype=torch.int64)
  fill = torch.empty(768 << 18, device=dev, dtype=torch.float32) if cold else None  # 768 MB: evicts L2 + Infinity Cache

  def go(dbgbuf=None):
    ops.gemm_nt(a, b, out, epi, m=dense, n_rows_dev=nrd, tile=tile, seed_dev=dbgbuf, **extra)

  for _ in range(3):
    go()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  side = torch.cuda.Stream()
  with torch.cuda.stream(side):
    go()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
      for _ in range(20):
        go()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  g.replay()
  s.record()
  g.replay()
  e.record()
  torch.cuda.synchronize()
  us = s.elapsed_time(e) / 20 * 1e3
  fl = 2.0 * live * N * K
  if tile not in (24, 25):  # (a gemm2 tile beside it: the time only)
    print('%-28s rows %d N %d K %d tile %d: %.1f us in a graph = %.0f TFLOP/s (%.3f of 2.5 PF)' %
          (name, live, N, K, tile, us, fl / us / 1e6, fl / us / 1e6 / 2500))
    return
  if cold:
    fill.fill_(1.0)
  go(dbg)
  torch.cuda.synchronize()
  d = dbg.cpu().double().view(256, 3, 8)
  print('%-28s rows %d N %d K %d tile %d: %.1f us in a graph = %.0f TFLOP/s (%.3f of 2.5 PF)%s' %
        (name, live, N, K, tile, us, fl / us / 1e6, fl / us / 1e6 / 2500, '   [budget launch: operands COLD]' if cold else ''))
  for n_tiles in sorted(set(d[:, 2, 6].long().tolist())):
    if n_tiles == 0:
      continue
    sel = d[:, 2, 6].long() == n_tiles
    blocks = int(sel.sum())
    c0, c1, pr = d[sel, 0].mean(0), d[sel, 1].mean(0), d[sel, 2].mean(0)
    S = pr[5].item()
    kt = S / n_tiles
    stage = 24576 if tile == 25 else 32768
    print('   %3d blocks with %d tiles (%d stages of %d KiB): whole block %6.0f cycles = %5.0f per tile, %4.0f per K-step' %
          (blocks, n_tiles, S, stage >> 10, pr[4].item(), pr[4].item() / n_tiles, pr[4].item() / S))
    for name_, c in (('consumer group 0', c0), ('consumer group 1', c1)):
      print('      %s: K-loop work %6.0f + barrier %6.0f | epilogue work %6.0f + barrier %6.0f | alive %6.0f' %
            (name_, c[0].item(), c[1].item(), c[2].item(), c[3].item(), c[4].item()))
    print('      producer wave 0  : prologue %5.0f | issue %6.0f (%4.0f per stage) + vmcnt wait %6.0f + barrier %6.0f | alive %6.0f   -> %4.1f B/clk/CU over its life' %
          (pr[0].item(), pr[1].item(), pr[1].item() / max(S - 2, 1), pr[2].item(), pr[3].item(), pr[4].item(),
           S * stage / max(pr[4].item(), 1)))
  sys.stdout.flush()


if __name__ == '__main__':
  live = int(sys.argv[1]) if len(sys.argv) > 1 else 3639
  cold = '--cold' in sys.argv
  if '--narrow' in sys.argv:  # the long-K GEMMs with narrow outputs: gemm2's phased 128x64 tile, gemm5 on 128x128 and 128x64 tiles
    if '--drop' in sys.argv:  # dropout on (as inside the step) against off
      for d in (0.0, 0.1):
        budget('FFN-down, dropout p = %.1f' % d, 6976, 6976, 512, 3072, 'BIAS_DROP_RES', tile=24, cold=cold, drop=d)
        budget('FFN-down, dropout p = %.1f' % d, live, 6976, 512, 3072, 'BIAS_DROP_RES', tile=25, cold=cold, drop=d)
      sys.exit(0)
    for rows in (live, 6976):
      for tile in (18, 24, 25):
        budget('FFN-down + bias + residual', rows, 6976, 512, 3072, 'BIAS_DROP_RES', tile=tile, cold=cold)
        budget('FFN-up input gradient + res', rows, 6976, 512, 3072, 'ADD_F32', tile=tile, cold=cold)
    sys.exit(0)
  budget('FFN-up + bias + GELU', live, 6976, 3072, 512, 'BIAS_GELU', cold=cold)
  budget('dGELU input gradient', live, 6976, 3072, 512, 'DGELU', cold=cold)
  budget('QKV + bias', live, 6976, 1536, 512, 'BIAS_BF16', cold=cold)
  budget('FFN-down + bias + residual', live, 6976, 512, 3072, 'BIAS_DROP_RES', cold=cold)
