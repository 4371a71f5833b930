"""How much of a GEMM's in-step duration is cache state?  The lab replays the same operands (warm in the 256 MB Infinity
Cache); in the step every operand was produced or last touched >= one kernel ago.  Variants, each a captured graph of
[prep ; GEMM] pairs, reported as pair time minus prep-only time:
  warm      no prep
  thrash    a 768 MB fill between launches (everything cold: HBM)
  producer  the FFN-up GEMM writes the A operand right before (what the step does), after a thrash"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
bf = torch.bfloat16
rows, DENSE, d, I = 3596, 6976, 512, 3072
R = ops.pad_rows(DENSE)
rnd = lambda *s, dtype=bf, scale=1.0: (torch.randn(*s, device=dev) * scale).to(dtype)
x, w1, w2 = rnd(R, d), rnd(I, d, scale=0.05), rnd(d, I, scale=0.05)
b1, b2 = rnd(I, dtype=torch.float32), rnd(d, dtype=torch.float32)
hpre, g = torch.zeros(R, I, device=dev, dtype=bf), torch.zeros(R, I, device=dev, dtype=bf)
res, z = rnd(R, d, dtype=torch.float32), torch.zeros(R, d, device=dev, dtype=torch.float32)
nrd = torch.tensor([rows], device=dev, dtype=torch.int32)
big = torch.zeros(768 << 20, device=dev, dtype=torch.uint8)

up = lambda: ops.gemm_nt(x, w1, g, 'BIAS_GELU', m=DENSE, bias=b1, out2=hpre, n_rows_dev=nrd, live_rows=rows)
down = lambda tile=0: ops.gemm_nt(g, w2, z, 'BIAS_DROP_RES', m=DENSE, bias=b2, res=res, drop_key=1, drop_p=0.1, n_rows_dev=nrd, tile=tile,
                                  live_rows=rows)
thrash = lambda: big.fill_(1)

# r06: `--pmc warm|cold|producer`: ONLY that variant of the FFN-down GEMM (the tile the step runs), 30 plain launches, for a
# rocprofv3 --pmc pass per variant (the L2 hit rate / fabric requests / request latency of the lab next to the step's:
# profiles/r06_pmc_cache_cold_lab.txt) -- one process per variant, because a profile groups launches by kernel name and grid.
if '--pmc' in sys.argv:
  mode = sys.argv[sys.argv.index('--pmc') + 1]
  for _ in range(30):
    if mode in ('cold', 'producer'):
      thrash()
    if mode == 'producer':
      up()
    down()
  torch.cuda.synchronize()
  sys.exit(0)


def graph_time(fn, iters=10):
  side = torch.cuda.Stream()
  with torch.cuda.stream(side):
    fn(); fn()
  torch.cuda.synchronize()
  gr = torch.cuda.CUDAGraph()
  with torch.cuda.graph(gr, stream=side):
    for _ in range(iters):
      fn()
  best = 1e9
  for _ in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr.replay(); s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / iters * 1e3)
  return best


t_thrash = graph_time(thrash)
t_up_cold = graph_time(lambda: (thrash(), up())) - t_thrash
print('thrash alone %.1f us; FFN-up after thrash %.1f us; FFN-up warm %.1f us' % (t_thrash, t_up_cold, graph_time(up)))
for tile in (0, 13, 18, 19):
  warm = graph_time(lambda: down(tile))
  cold = graph_time(lambda: (thrash(), down(tile))) - t_thrash
  prod = graph_time(lambda: (thrash(), up(), down(tile))) - t_thrash - t_up_cold
  print('FFN-down tile %2d: warm %.1f us   all-cold %.1f us   A fresh from FFN-up (rest cold) %.1f us' % (tile, warm, cold, prod))
