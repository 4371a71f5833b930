"""One replayed training step (between two minibatch copies) from a rocprofv3 kernel-trace database, in launch order,
with per-kernel duration and the gap to the previous kernel:  python tools/graph_sequence.py <results.db> [which]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute('select name, start, end, grid_x, workgroup_x from kernels order by start').fetchall()
idx = [i for i, r in enumerate(rows) if 'copyBuffer' in r[0] and r[3] // max(r[4], 1) == 256]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
mid, nxt = idx[which], idx[which + 1]
prev_end, tot, gaps = None, 0.0, 0.0
for name, st, en, gx, wx in rows[mid:nxt + 1]:
  gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
  name = re.sub(r'\(.*$', '', name) if not name.startswith('void at::') else name
  print('%8.1f us  gap %6.1f  %5d x %4d  %s' % ((en - st) / 1e3, gap, gx // max(wx, 1), wx, name[:80]))
  tot += (en - st) / 1e3
  gaps += max(gap, 0.0)
  prev_end = max(en, prev_end or 0)
print('%d kernels, kernel time %.1f us, gaps %.1f us, span %.1f us' % (nxt - mid, tot, gaps, (rows[nxt][1] - rows[mid][1]) / 1e3))
