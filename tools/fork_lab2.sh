cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fork_lab2
mkdir -p $O
cd $R
timeout 400 python tools/fork_diag.py 1 4 16 32 64 21 > $O/diag.txt 2>&1
cat $O/diag.txt | tail -20
for rep in 1 2; do
for f in 0 32 96 100; do
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
