"""Phase timestamps of the attention backward blocks at the headline shape (lab build: python -m mmt_amd.build --instr;
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so python tools/attn_budget.py).  Packed rows with the synthetic MSRVTT valid
lengths (7 experts x U{0..30} tokens + 8), B = 32, 4 heads x 128."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmt_amd import _lib, ops  # noqa: E402
from mmt_amd.ops import _p, _stream  # noqa: E402

dev = torch.device('cuda:0')
H, d, B, S = 4, 512, 32, 218
rs = np.random.RandomState(0)
lens = 8 + rs.randint(0, 31, size=(B, 7)).sum(1)
if '--dense' in sys.argv:
  lens[:] = S
cu_h = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
rows = int(cu_h[-1])
R = ops.pad_rows(B * S)
qkv = (torch.randn(R, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
mask = torch.zeros(R, device=dev)
cu = torch.from_numpy(cu_h).to(dev)
ctx = torch.zeros(R, d, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(R, H, device=dev)
dctx = (torch.randn(R, d, device=dev) * 0.1).to(torch.bfloat16)
dqkv = torch.zeros_like(qkv)
delta = torch.zeros(R, d // 64, device=dev)
thr, sc = ops.dropout_params(0.1)
L = _lib.lib()
Ld = ctypes.CDLL(_lib.LIB_PATH)
scale = 128 ** -0.5


def fwd():
  _lib.check(L.mmt_attn_fwd(_p(qkv), _p(cu), _p(mask), _p(ctx), _p(lse), B, S, H, d, scale, 7, thr, sc, None, None, _stream()), 'f')


SCHED = '--nosched' not in sys.argv  # the engine's block order (attn_sched.h); --nosched: slot order of the plain API
work = torch.empty(L.mmt_attn_schedule_words(B, S, H), device=dev, dtype=torch.int32)
_lib.check(L.mmt_attn_schedule(_p(cu), B, S, H, _p(work), _stream()), 'sched')
print('block order:', 'scheduled (longest first, dead slots last)' if SCHED else 'slot order (dK/dV tiles, then dQ tiles)')


def bwd():  # (the delta sums are in place after the first plain call: the kernel alone, as the engine launches it)
  _lib.check(L.mmt_attn_bwd_ex(_p(qkv), _p(cu), _p(mask), _p(ctx), _p(lse), _p(dctx), _p(dqkv), _p(delta), 1, B, S, H, d,
                               scale, 7, thr, sc, None, None, _p(work) if SCHED else None, _stream()), 'b')


fwd()
_lib.check(L.mmt_attn_bwd(_p(qkv), _p(cu), _p(mask), _p(ctx), _p(lse), _p(dctx), _p(dqkv), _p(delta), B, S, H, d, scale,
                          7, thr, sc, None, None, _stream()), 'b0')
for _ in range(3):
  bwd()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
  bwd()
e.record()
torch.cuda.synchronize()
print('attention backward, %d live rows (lengths %d..%d): %.1f us per launch (eager, back to back)' %
      (rows, lens.min(), lens.max(), s.elapsed_time(e) / 20 * 1e3))
nblk = 8 * H * B
if '--fwd' in sys.argv:
  bwd = fwd
dbg = torch.zeros(nblk, 16, device=dev, dtype=torch.int64)
Ld.mmt_debug_set_attn_buffer(ctypes.c_void_p(dbg.data_ptr()))
bwd()
torch.cuda.synchronize()
x = dbg.cpu().double()
ran = x[:, 0] > 0
xcc = x[:, 9].long() & 0xf
# clock64 is CU-local: phase durations only.  Block end on the chip-wide 100 MHz counter (x[:, 11]); start = end - duration.
dur_all = x[:, 4] - x[:, 0]
for k in (1, 2, 3, 4, 8):
  x[:, k] = torch.where(x[:, k] > 0, x[:, k] - x[:, 0], x[:, k])
wend = (x[:, 11] - x[ran, 11].min()) * 24.0   # ~2.4 GHz shader cycles per 10 ns tick (approximate)
x[:, 0] = wend - dur_all
x[:, 0] -= x[ran, 0].min()
for k in (1, 2, 3, 4, 8):
  x[:, k] = torch.where(x[:, k] > 0, x[:, k] + x[:, 0], x[:, k])
live = ran & (x[:, 5] > 0)
dead = ran & ~live
print('%d blocks: %d live, %d exit at once (no tile for them); kernel span (max over XCCs) %.0f cycles' %
      (nblk, int(live.sum()), int(dead.sum()), x[ran, 4].max().item()))
print('live blocks per XCC: ' + ' '.join('%d:%d' % (c, int((live & (xcc == c)).sum())) for c in range(8)) +
      ' | last exit per XCC (cycles): ' + ' '.join('%d:%.0f' % (c, x[ran & (xcc == c), 4].max().item()) for c in range(8)
                                                     if (ran & (xcc == c)).any()))
for role, name in ((0, 'dQ   '), (1, 'dK/dV'), (2, 'fwd  ')):
  for it in (1, 2, 3, 4):
    y = x[live & (x[:, 6] == role) & (x[:, 5] == it)]
    if not len(y):
      continue
    second = (y[:, 8] - y[:, 2]).mean().item() if it > 1 else float('nan')
    print('%s blocks with %d tile iterations: %3d | start %6.0f | prologue (entry -> loads issued) %5.0f | '
          'first tile landed after %5.0f more | iteration 1 (compute + wait for tile 2) %5.0f | loop total %6.0f | '
          'epilogue (transposes + stores + drain) %5.0f | block %6.0f cycles'
          % (name, it, len(y), y[:, 0].mean().item(), (y[:, 1] - y[:, 0]).mean().item(), (y[:, 2] - y[:, 1]).mean().item(),
             second, (y[:, 3] - y[:, 2]).mean().item(), (y[:, 4] - y[:, 3]).mean().item(), (y[:, 4] - y[:, 0]).mean().item()))
y = x[live]
cuid = (y[:, 9].long() & 0xf) * 65536 + (y[:, 10].long() & 0xff00)
cus = cuid.unique()
per = torch.stack([(cuid == c).sum() for c in cus]).double()
print('live blocks ran on %d CUs: %.2f per CU (max %d); end of the last block per CU: mean %.0f, max %.0f cycles; '
      'start of the LAST-starting live block %.0f' %
      (len(cus), per.mean().item(), int(per.max().item()),
       torch.stack([(y[cuid == c, 4]).max() for c in cus]).mean().item(), y[:, 4].max().item(), y[:, 0].max().item()))
st = (y[:, 0] / 4000.0).floor()
print('live-block start times, bins of 4k cycles: ' + ' '.join('%d:%d' % (int(v) * 4, int((st == v).sum())) for v in st.unique()))
