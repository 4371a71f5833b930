"""Cycle budget of the FFN-sized GEMMs of the headline step, per phase.  Needs the lab build:
   python -m mmt_amd.build --instr
   MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so python tools/gemm2_budget.py [rows]
For each of the four shapes (FFN-up + GELU, its dGELU input gradient, FFN-down + dropout + residual, its input gradient)
launched as the step launches them (dense-sized grid, live row count on the device, the tile dispatch_tile picks), wave 0
of every live block reports s_memtime deltas: prologue (entry -> first K-step), the K-loop split into counted-vmcnt wait /
barrier / LDS-DMA issue / LDS-read + MFMA, the epilogue split into head (second-operand prefetch + barrier), accumulator
staging through LDS, the row sweep (VALU + global stores), and the store drain; plus where and when the block ran, from
which the tail-round idle time of the CUs is derived."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16


def rnd(*shape, dtype=bf, scale=1.0):
  return (torch.randn(*shape, device=dev) * scale).to(dtype)


def budget(name, live, dense, N, K, epi, tile=0, **kw):
  R = ops.pad_rows(dense)
  a, b = rnd(R, K), rnd(N, K, scale=0.05)
  out = torch.zeros(R, N, device=dev, dtype=torch.float32 if epi in ('BIAS_DROP_RES', 'ADD_F32', 'F32') else bf)
  nrd = torch.tensor([live], device=dev, dtype=torch.int32)
  extra = {}
  if epi in ('BIAS_GELU', 'BIAS_DROP_RES'):
    extra['bias'] = rnd(N, dtype=torch.float32)
  if epi == 'BIAS_GELU':
    extra['out2'] = torch.zeros(R, N, device=dev, dtype=bf)
  if epi in ('BIAS_DROP_RES', 'ADD_F32'):
    extra['res'] = rnd(R, N, dtype=torch.float32)
  if epi == 'DGELU':
    extra['aux'] = rnd(R, N)
  dbg = torch.zeros(8192, 16, device=dev, dtype=torch.int64)

  def go(dbgbuf):
    ops.gemm_nt(a, b, out, epi, m=dense, n_rows_dev=nrd, tile=tile, seed_dev=dbgbuf, **extra)

  for _ in range(3):
    go(dbg)
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  g = torch.cuda.CUDAGraph()
  side = torch.cuda.Stream()
  with torch.cuda.stream(side):
    go(dbg)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
      for _ in range(20):
        go(dbg)
  g.replay()
  s.record()
  g.replay()
  e.record()
  torch.cuda.synchronize()
  us = s.elapsed_time(e) / 20 * 1e3
  dbg.zero_()
  go(dbg)
  torch.cuda.synchronize()
  d = dbg.cpu().double()
  d = d[d[:, 7] > 0]
  kt = d[0, 7].item()
  m = d.mean(0)
  # s_memtime (clock64) is local to a CU's clock domain: durations only.  Start / end of a block on the chip-wide 100 MHz
  # counter (wall_clock64), converted to shader cycles with the measured ratio of the two over whole blocks.
  xcc = d[:, 14].long() & 0xf
  w_entry, w_end = d[:, 12] - d[:, 12].min(), d[:, 13] - d[:, 12].min()
  ratio = (d[:, 6].sum() / (w_end - w_entry).sum()).item()   # shader cycles per 10 ns tick
  t_entry, t_end = w_entry * ratio, w_end * ratio
  span = t_end.max().item()
  dur = d[:, 6]
  cu = xcc * 65536 + (d[:, 15].long() & 0xff00)  # (xcc, se, sh, cu)
  cus = cu.unique()
  last_end = torch.stack([t_end[cu == c].max() for c in cus])
  first_start = torch.stack([t_entry[cu == c].min() for c in cus])
  per_cu = torch.stack([(cu == c).sum() for c in cus]).double()
  busy = torch.stack([dur[cu == c].sum() for c in cus])
  tail_idle = (span - last_end).mean().item()
  head_idle = first_start.mean().item()
  ncu = len(cus)
  flops = 2.0 * live * N * K
  print('%s  rows %d (grid for %d)  N %d K %d  tile %d: %.1f us in a graph = %.0f TFLOP/s (%.3f of 2.5 PF); %d live blocks on %d CUs '
        '(%.2f per CU, max %d)' % (name, live, dense, N, K, tile, us, flops / us / 1e6, flops / us / 1e6 / 2500, d.shape[0], ncu,
                                   per_cu.mean().item(), int(per_cu.max().item())))
  print('   block (wave 0, mean cycles): prologue %5.0f | K-loop %6.0f = %d K-steps x (wait %4.0f + barrier %4.0f + issue %4.0f + '
        'read/MFMA %4.0f) | epilogue %5.0f = head %4.0f + LDS staging %4.0f + sweep %5.0f + store drain %4.0f | whole block %6.0f'
        % (m[8], m[4], kt, m[0] / kt, m[1] / kt, m[2] / kt, m[3] / kt, (d[:, 6] - d[:, 8] - d[:, 4]).mean().item(), m[9], m[10], m[11],
           (d[:, 6] - d[:, 8] - d[:, 4] - d[:, 5]).mean().item(), dur.mean().item()))
  print('   launch: span (first entry -> last exit) %6.0f cycles; %d of 256 CUs got work, %.2f blocks resident per such CU on '
        'average over the span; mean idle per CU before its first block %5.0f, after its last block (tail round) %5.0f cycles; '
        'MFMA-only time of this GEMM %5.0f cycles per CU; shader clock %.2f GHz'
        % (span, ncu, busy.sum().item() / (span * ncu), head_idle, tail_idle, flops / 256 / 4096.0, ratio / 10.0))
  return us


nums = [a for a in sys.argv[1:] if a.isdigit()]
rows = int(nums[0]) if nums else 3639
dense = 6976
print('# gemm2 cycle budget, lab build (s_memtime ticks of wave 0; the ticks themselves inflate a loop by ~10-20 %)')
budget('FFN-up   + bias + erf-GELU (2 outputs) ', rows, dense, 3072, 512, 'BIAS_GELU')
budget('dGELU input gradient (reads pre-act)   ', rows, dense, 3072, 512, 'DGELU')
budget('FFN-down + bias + dropout + residual   ', rows, dense, 512, 3072, 'BIAS_DROP_RES')
budget('FFN-up input gradient + residual grad  ', rows, dense, 512, 3072, 'ADD_F32')
if '--alt' in sys.argv:
  for t in (11, 3, 15, 16):
    budget('FFN-up on tile %d' % t, rows, dense, 3072, 512, 'BIAS_GELU', tile=t)
  for t in (13, 18):
    budget('FFN-down on tile %d' % t, rows, dense, 512, 3072, 'BIAS_DROP_RES', tile=t)
