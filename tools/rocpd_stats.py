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
  args = ap.parse_args()
  c = sqlite3.connect(args.db)
  cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
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
