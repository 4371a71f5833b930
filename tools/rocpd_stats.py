"""Per-kernel stats (calls, total/avg/min/max ns, %) from a rocprofv3 rocpd sqlite database.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--csv out.csv] [--top 40] [--skip-first N]

rocprofv3 on this image writes `*_results.db` by default; this is the `--stats` summary of that file.
"""
import argparse
import csv
import re
import sqlite3
import sys


def short(name):
  name = re.sub(r'\(.*$', '', name) if not name.startswith('void at::') else name
  return name[:110]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('db')
  ap.add_argument('--csv')
  ap.add_argument('--top', type=int, default=50)
  ap.add_argument('--by-grid', action='store_true', help='separate rows per launch geometry')
  ap.add_argument('--sequence', type=int, default=0, help='print the last N dispatches in order with gaps')
  args = ap.parse_args()
  c = sqlite3.connect(args.db)
  cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
  if args.sequence:
    seq = c.execute('select name, start, end, grid_x, workgroup_x, stream_id from kernels order by start').fetchall()[-args.sequence:]
    prev_end = None
    for name, st, en, gx, wx, sid in seq:
      gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
      print('%8.1f us  gap %7.1f  s%-3s %5d x %4d  %s' % ((en - st) / 1e3, gap, sid, gx // max(wx, 1), wx, short(name)[:70]))
      prev_end = max(en, prev_end or 0)
    return
  if args.by_grid:
    rows = [('%s [%d x %d]' % (short(n), gx // max(wx, 1), wx), d) for n, d, gx, wx in
            c.execute('select name, (end - start), grid_x, workgroup_x from kernels')]
  else:
    rows = c.execute('select name, (end - start) from kernels').fetchall()
  agg = {}
  for name, dur in rows:
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
  total = sum(a[1] for a in agg.values())
  out = sorted(agg.items(), key=lambda kv: -kv[1][1])
  if args.csv:
    with open(args.csv, 'w', newline='') as f:
      w = csv.writer(f)
      w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
      for name, a in out:
        w.writerow([name, a[0], a[1], '%.1f' % (a[1] / a[0]), '%.3f' % (100.0 * a[1] / total), a[2], a[3]])
  print('total kernel time %.3f ms over %d dispatches, %d distinct kernels' % (total / 1e6, len(rows), len(agg)))
  for name, a in out[:args.top]:
    print('%6.2f%% %7d calls  avg %9.1f ns  %s' % (100.0 * a[1] / total, a[0], a[1] / a[0], short(name)))


if __name__ == '__main__':
  main()
