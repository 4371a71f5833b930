"""gemm_ln.hip lab: difference against GEMM + LayerNorm launches, and timings (graph of 20 launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops
dev = torch.device('cuda:0')
M, K, N = 3639, 512, 512
R = ops.pad_rows(6976)
g = torch.Generator(device='cpu').manual_seed(1)
a = (torch.randn(R, K, generator=g)).to(torch.bfloat16).to(dev)
w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
bias, res = torch.randn(N, generator=g).to(dev), torch.randn(R, N, generator=g).to(dev)
gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
nrd = torch.tensor([M], device=dev, dtype=torch.int32)
z, h32, h16, mean, rstd = ops.gemm_nt_ln_fwd(a, w, bias, res, gamma, beta, 1e-12, m=6976, n_rows_dev=nrd)
for tile in (13, 18):
  z2 = torch.zeros(R, N, device=dev)
  ops.gemm_nt(a, w, z2, 'BIAS_DROP_RES', m=6976, bias=bias, res=res, n_rows_dev=nrd, tile=tile)
  g32, g16, gm, gr = ops.ln_fwd(z2, gamma, beta, 1e-12, rows=6976, n_rows_dev=nrd)
  print('tile', tile, 'z max diff', (z[:M] - z2[:M]).abs().max().item(), 'equal', torch.equal(z[:M], z2[:M]), 'mean', (mean[:M] - gm[:M]).abs().max().item(),
        'h32', (h32[:M] - g32[:M]).abs().max().item())
ref = a[:M].float() @ w.float().t() + bias + res[:M]
print('vs torch fp32: z', (z[:M] - ref).abs().max().item())

def timeit(f, iters=20):
  side = torch.cuda.Stream()
  with torch.cuda.stream(side):
    f(); f()
  torch.cuda.synchronize()
  gr = torch.cuda.CUDAGraph()
  with torch.cuda.graph(gr, stream=side):
    for _ in range(iters):
      f()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  gr.replay(); s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / iters * 1e3
zz = torch.zeros(R, N, device=dev)
print('fused           %.1f us' % timeit(lambda: ops.gemm_nt_ln_fwd(a, w, bias, res, gamma, beta, 1e-12, m=6976, n_rows_dev=nrd)))
def two():
  ops.gemm_nt(a, w, zz, 'BIAS_DROP_RES', m=6976, bias=bias, res=res, n_rows_dev=nrd)
  ops.ln_fwd(zz, gamma, beta, 1e-12, rows=6976, n_rows_dev=nrd)
print('gemm + ln_fwd   %.1f us (incl. python-side allocations of ln_fwd outputs)' % timeit(two))
print('gemm alone      %.1f us' % timeit(lambda: ops.gemm_nt(a, w, zz, 'BIAS_DROP_RES', m=6976, bias=bias, res=res, n_rows_dev=nrd)))
