"""Per-kernel averages of rocprofv3 --pmc counters from rocpd sqlite databases (one per pass).

    python tools/rocpd_pmc.py pass1_results.db [pass2_results.db ...] [--csv out.csv] [--match substr]

Prints, per kernel name: calls, avg duration, and the average of every collected counter per dispatch.
Run it ON the GPU box (the databases are too large to pull back) and keep only the CSV.
"""
import argparse
import csv
import re
import sqlite3


def short(name):
  return re.sub(r'\(.*$', '', name)[:70] if not name.startswith('void at::') else name[:70]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('dbs', nargs='+')
  ap.add_argument('--csv')
  ap.add_argument('--match', default='')
  ap.add_argument('--top', type=int, default=30)
  args = ap.parse_args()
  agg = {}    # name -> {counter: [sum, n]}
  dur = {}    # name -> [sum_ns, n]
  counters = []
  for db in args.dbs:
    c = sqlite3.connect(db)
    q = ('select kernel_name, counter_name, value, (end - start), grid_size, workgroup_size from counters_collection')
    for name, cn, val, d, grid, wg in c.execute(q):
      if args.match and args.match not in name:
        continue
      key = '%s [grid %d x %d]' % (short(name), grid // max(wg, 1), wg)
      a = agg.setdefault(key, {})
      s = a.setdefault(cn, [0.0, 0])
      s[0] += val; s[1] += 1
      if cn not in counters:
        counters.append(cn)
      dd = dur.setdefault((key, cn), [0, 0])
      dd[0] += d; dd[1] += 1
  rows = []
  for key, a in agg.items():
    first = next(iter(a))
    dsum, dn = dur[(key, first)]
    rows.append((dsum, key, dn, dsum / dn, {cn: v[0] / v[1] for cn, v in a.items()}))
  rows.sort(reverse=True)
  if args.csv:
    with open(args.csv, 'w', newline='') as f:
      w = csv.writer(f)
      w.writerow(['kernel', 'dispatches', 'avg_ns'] + counters)
      for _, key, n, avg, cv in rows:
        w.writerow([key, n, '%.0f' % avg] + ['%.0f' % cv.get(cn, float('nan')) for cn in counters])
  for _, key, n, avg, cv in rows[:args.top]:
    print('%-90s n=%5d avg %8.1f us  ' % (key, n, avg / 1e3) + ' '.join('%s=%.3g' % (cn, cv[cn]) for cn in counters if cn in cv))


if __name__ == '__main__':
  main()
