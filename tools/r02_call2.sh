cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c2
mkdir -p $O
for v in alone chain chain_live; do
  rm -rf /tmp/p_$v
  timeout 300 rocprofv3 --kernel-trace -d /tmp/p_$v -o t -- python $R/tools/chain_lab.py --variant $v > $O/chain_$v.log 2>&1
  DB=$(find /tmp/p_$v -name "*.db" | head -1)
  echo "== $v" >> $O/chain.txt
  tail -1 $O/chain_$v.log >> $O/chain.txt
  python $R/tools/rocpd_stats.py $DB --by-grid --top 6 >> $O/chain.txt 2>&1
done
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof -o step -- python $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python - <<PY > $O/graph_sequence.txt 2>&1
import sqlite3, re
c = sqlite3.connect("$DB")
rows = c.execute('select name, start, end, grid_x, workgroup_x from kernels order by start').fetchall()
# a replayed step in the middle of the timed region: between two big minibatch copies
idx = [i for i, r in enumerate(rows) if 'copyBuffer' in r[0] and r[3] // max(r[4], 1) == 256]
mid = idx[len(idx) // 2]
nxt = idx[len(idx) // 2 + 1]
prev_end = None
tot = gaps = 0.0
for name, st, en, gx, wx in rows[mid:nxt + 1]:
  gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
  name = re.sub(r'\(.*$', '', name) if not name.startswith('void at::') else name
  print('%8.1f us  gap %6.1f  %5d x %4d  %s' % ((en - st) / 1e3, gap, gx // max(wx, 1), wx, name[:80]))
  tot += (en - st) / 1e3; gaps += max(gap, 0.0)
  prev_end = max(en, prev_end or 0)
print('kernel time %.1f us, gaps %.1f us, span %.1f us' % (tot, gaps, (rows[nxt][1] - rows[mid][1]) / 1e3))
PY
ls $O
