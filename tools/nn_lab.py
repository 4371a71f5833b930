"""Input-gradient GEMMs: NN form (weight as stored, transpose reads) against the NT form on a transposed weight copy.
python tools/nn_lab.py [text]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402
from tools.gemm_lab import rnd, timeit  # noqa: E402

dev = torch.device('cuda:0')
text = len(sys.argv) > 1 and sys.argv[1] == 'text'
d, I = (768, 3072) if text else (512, 3072)
for rows in ((560, 960) if text else (3573, 6976)):
  R = ops.pad_rows(rows)
  nr = torch.tensor([rows], device=dev, dtype=torch.int32)
  for (N, K, epi) in ((I, d, 'DGELU'), (d, I, 'ADD_F32'), (d, d, 'BF16'), (d, 3 * d, 'ADD_F32')):
    a = rnd(R, K)
    w = rnd(K, N, scale=0.05)       # [K, N]: the weight as stored ([out, in] with out = K)
    wt = w.t().contiguous()         # [N, K]: the transposed shadow the NT form needs
    res, aux = rnd(R, N, dtype=torch.float32), rnd(R, N)
    out = torch.zeros(R, N, device=dev, dtype=torch.float32 if epi == 'ADD_F32' else torch.bfloat16)
    fns = [lambda: ops.gemm_nt(a, wt, out, epi, m=R, res=res, aux=aux, n_rows_dev=nr),
           lambda: ops.gemm_nn(a, w, out, epi, m=R, res=res, aux=aux, n_rows_dev=nr),
           lambda: ops.gemm_nn(a, w, out, epi, m=R, res=res, aux=aux, n_rows_dev=nr, tile=13),
           lambda: ops.gemm_nn(a, w, out, epi, m=R, res=res, aux=aux, n_rows_dev=nr, tile=14) if N % 128 == 0 else None]
    torch.cuda.synchronize()
    ts = timeit(fns)
    print('rows %5d N %4d K %4d %-8s | NT auto %6.1f  NN auto %6.1f  NN t13 %6.1f  NN t14 %6.1f' % ((rows, N, K, epi) + tuple(ts)))
