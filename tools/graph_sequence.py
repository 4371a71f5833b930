"""One replayed training step (from its first text-head kernel to the first kernel of the next step) from a
rocprofv3 kernel-trace database, in start order,
with per-kernel start offset, duration, the gap to the latest end seen so far (negative = it OVERLAPS an earlier kernel:
a parallel branch of the step graph) and the queue it ran on:

    python tools/graph_sequence.py <results.db> [which]

Footer: sum of kernel durations vs the span of the step and the time at least one kernel was running (union)."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else '0')
rows = c.execute('select name, start, end, grid_x, workgroup_x, %s from kernels order by start' % qcol).fetchall()
# A step OPENS with the first text-head kernel (th_fwd1_kernel: once per step in every mode; the video plan kernel when
# the text heads run elsewhere).  (Up to r02 every step opened with the device-to-device copy of the minibatch; with
# input slots there is no such copy.)
starts = [i for i, r in enumerate(rows) if 'th_fwd1' in r[0]] or [i for i, r in enumerate(rows) if 'video_plan_kernel' in r[0]]
if len(starts) < 3:
  sys.exit('fewer than three steps in the trace')
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
mid, nxt = starts[which], starts[which + 1]
t0 = rows[mid][1]
prev_end, tot, union, overlapped = None, 0.0, 0.0, 0
queues = {}
for name, st, en, gx, wx, q in rows[mid:nxt + 1]:
  gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
  name = re.sub(r'\(.*$', '', name) if not name.startswith('void at::') else name
  qi = queues.setdefault(q, len(queues))
  print('%8.1f  %7.1f us  gap %7.1f  q%d %5d x %4d  %s' % ((st - t0) / 1e3, (en - st) / 1e3, gap, qi, gx // max(wx, 1), wx, name[:80]))
  tot += (en - st) / 1e3
  if prev_end is None or st >= prev_end:
    union += (en - st) / 1e3
  else:
    overlapped += 1
    union += max(0.0, (en - prev_end) / 1e3)
  prev_end = max(en, prev_end or 0)
span = (rows[nxt][1] - rows[mid][1]) / 1e3
print('%d kernels on %d queues, sum of durations %.1f us, span %.1f us, busy (union) %.1f us, %d kernels start before an '
      'earlier one has ended' % (nxt - mid, len(queues), tot, span, union, overlapped))
