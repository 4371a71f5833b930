"""Why do the N = 512 GEMMs run ~40 % slower inside the step than alone?  Variants of the FFN pair under one HIP graph,
timed per kernel by rocprofv3 (run this script under `rocprofv3 --kernel-trace`, one variant per process):
   python tools/chain_lab.py --variant alone|chain|chain_live|chain_live_ln [--iters 30]
alone      : FFN-down repeated on a static A operand (what tools/gemm_lab.py measures)
chain      : FFN-up -> FFN-down, A freshly produced by the previous kernel
chain_live : the same with dense-sized buffers, device live-row count and row_index (token packing)
chain_live_ln : ... + LayerNorm after it (the real layer tail)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--variant', default='alone')
ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--live', type=int, default=3583)
ap.add_argument('--dense', type=int, default=6976)
args = ap.parse_args()
dev = torch.device('cuda:0')
bf = torch.bfloat16
d, I = 512, 3072
live, dense = args.live, args.dense
packed = args.variant.startswith('chain_live')
M = dense if packed else live
R = ops.pad_rows(M)


def rnd(*shape, dtype=bf, scale=1.0):
  return (torch.randn(*shape, device=dev) * scale).to(dtype)


a16 = rnd(R, d)
w1, w2 = rnd(I, d, scale=0.05), rnd(d, I, scale=0.05)
b1, b2 = rnd(I, dtype=torch.float32), rnd(d, dtype=torch.float32)
hpre, g = torch.zeros(R, I, device=dev, dtype=bf), rnd(R, I)
res = rnd(R, d, dtype=torch.float32)
z2 = torch.zeros(R, d, device=dev)
h32, gam, bet = torch.zeros(R, d, device=dev), torch.ones(d, device=dev), torch.zeros(d, device=dev)
nr = torch.tensor([live], device=dev, dtype=torch.int32) if packed else None
ridx = torch.arange(R, device=dev, dtype=torch.int32) if packed else None
seed = torch.tensor([5], device=dev, dtype=torch.int32)


def up():
  ops.gemm_nt(a16, w1, hpre, 'BIAS_GELU', m=M, bias=b1, out2=g, n_rows_dev=nr)


def down():
  ops.gemm_nt(g, w2, z2, 'BIAS_DROP_RES', m=M, bias=b2, res=res, drop_key=7, drop_p=0.1, n_rows_dev=nr, row_index=ridx,
              seed_dev=seed)


def body():
  if args.variant == 'alone':
    down()
  else:
    up()
    down()
    if args.variant.endswith('_ln'):
      ops.ln_fwd(z2, gam, bet, 1e-12, rows=M)


side = torch.cuda.Stream()
with torch.cuda.stream(side):
  for _ in range(3):
    body()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=side):
  for _ in range(args.iters):
    body()
for _ in range(3):
  gr.replay()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
gr.replay()
e.record()
torch.cuda.synchronize()
print('%s: %.1f us per iteration' % (args.variant, s.elapsed_time(e) / args.iters * 1e3))
