"""How fast does the vendor library (torch.matmul -> hipBLASLt/rocBLAS) run the step's GEMM shapes?  Only a yardstick
for tools/gemm_lab.py numbers (plain GEMM, no fused epilogue); the product never calls it."""
import torch

def bench(m, n, k, iters=20, reps=5):
  a = torch.randn(m, k, device='cuda', dtype=torch.bfloat16)
  w = torch.randn(n, k, device='cuda', dtype=torch.bfloat16)
  out = torch.empty(m, n, device='cuda', dtype=torch.bfloat16)
  s = torch.cuda.Stream()
  with torch.cuda.stream(s):
    for _ in range(3):
      torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
      for _ in range(iters):
        torch.matmul(a, w.t(), out=out)
    best = 1e9
    for _ in range(reps):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
      best = min(best, e0.elapsed_time(e1) / iters)
  print('M %5d N %5d K %5d : %7.2f us  %7.1f TFLOP/s' % (m, n, k, best * 1e3, 2.0 * m * n * k / best / 1e9))

import sys
if len(sys.argv) > 1 and sys.argv[1] == 'big':  # configs[4] (d1024 L6 I6144, batch 128: ~22.8k live rows of 55.9k) and configs[3]
  for m, n, k in ((22784, 6144, 1024), (22784, 1024, 6144), (22784, 3072, 1024), (22784, 1024, 1024), (22784, 1024, 3072),
                  (11904, 3072, 512), (11904, 512, 3072), (8192, 8192, 8192)):
    bench(m, n, k)
else:
  for m in (3573, 6976):
    for n, k in ((3072, 512), (512, 3072), (1536, 512), (512, 512), (512, 1536)):
      bench(m, n, k)
