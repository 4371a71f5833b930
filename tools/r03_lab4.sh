cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_lab4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_eval_loop_gpu.py -q -m gpu -s -x 2>&1 | grep -v "^$" | cut -c1-300 | head -80 > $O/pytest_eval.txt
cat $O/pytest_eval.txt | tail -50
for cfg in 3 4; do
for sp in 1 0; do
  MMT_ATTN_BWD_SPLIT=$sp timeout 300 python bench.py --config $cfg --steps 40 --warmup 8 --no-dense 2>$O/err_ab.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); print('config $cfg attn bwd split=$sp  %.4f ms/step  %.0f pairs/s' % (d['ms_per_step'], d['value']))
" >> $O/ab.txt
done
done
cat $O/ab.txt
timeout 600 python tools/shape_bench.py 2>&1 | tail -5
