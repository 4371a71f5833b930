# r03 lab: the step graph with parallel branches (GraphedTrainStep fork bits) -- same-box A/B of the bit combinations,
# then a kernel trace of the default.   gpurun -- 'bash tools/fork_lab.sh'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fork_lab
mkdir -p $O
cd $R
for rep in 1 2; do
for f in 0 16 48 1 5 21 117 119 3; do
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-dense --fork $f 2>$O/err_$f.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
  try:
    d = json.loads(l); print('fork %3s  %.4f ms/step  %.0f pairs/s  loss %.6f -> %.6f' % ('$f', d['ms_per_step'], d['value'], d['first_loss'], d['final_loss']))
  except Exception as e: print('fork $f failed', l[:200])
" >> $O/ab.txt
done
done
cat $O/ab.txt
cd /tmp
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o step -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-dense > $O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/tools/graph_sequence.py $DB > $O/graph_sequence_fork.txt 2>&1
python $R/tools/rocpd_stats.py $DB --by-grid --top 60 > $O/kernel_stats_fork_by_grid.txt 2>&1
tail -3 $O/graph_sequence_fork.txt
