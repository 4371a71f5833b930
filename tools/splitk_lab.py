"""Split-K lab for the N=512 GEMMs of the step (FFN down-projection K=3072, input-gradient K=1536/3072): partial-slab
kernel alone and with the reduce+epilogue kernel, against the single-pass tiles.  python tools/splitk_lab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import ops  # noqa: E402
from tools.gemm_lab import rnd, timeit  # noqa: E402

dev = torch.device('cuda:0')


def main():
  text = len(sys.argv) > 1 and sys.argv[1] == 'text'  # the text tower's shapes: 960 (or packed ~560) rows, d = 768
  for rows in ((960, 560) if text else (3573, 6976)):
    R = ops.pad_rows(rows)
    for (N, K) in (((768, 3072), (768, 2304), (768, 768)) if text else ((512, 3072), (512, 1536), (512, 512), (1536, 512))):
      a, b = rnd(R, K), rnd(N, K, scale=0.05)
      res = rnd(R, N, dtype=torch.float32)
      out = torch.zeros(R, N, device=dev)
      ws = torch.empty(16 * (R + 128) * N, device=dev)
      ref = a[:rows].float() @ b.float().t() + res[:rows]
      names, fns = [], []
      for tile in (0, 13, 18):
        names.append('tile%d' % tile)
        fns.append(lambda tile=tile: ops.gemm_nt(a, b, out, 'ADD_F32', m=rows, res=res, tile=tile))
      for wide in (0, 1, 2):
        for splits in (2, 3, 4, 6):
          if K // 64 < splits:
            continue
          ops.gemm_nt_splitk(a, b, out, 'ADD_F32', m=rows, res=res, splits=splits, wide=wide, ws=ws)
          err = (out[:rows] - ref).abs().max().item()
          assert err < 2e-2, (wide, splits, err)
          names.append('%s s%d part' % ('nwW'[wide], splits))
          fns.append(lambda s=splits, w=wide: ops.gemm_nt_splitk(a, b, out, 'ADD_F32', m=rows, res=res, splits=s, wide=w,
                                                                 ws=ws, no_epilogue=True))
          names.append('%s s%d full' % ('nwW'[wide], splits))
          fns.append(lambda s=splits, w=wide: ops.gemm_nt_splitk(a, b, out, 'ADD_F32', m=rows, res=res, splits=s, wide=w, ws=ws))
      torch.cuda.synchronize()
      ts = timeit(fns)
      print('rows %5d N %4d K %4d | ' % (rows, N, K) + '  '.join('%s %5.1f' % (n, t) for n, t in zip(names, ts)))


if __name__ == '__main__':
  main()
