cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_lab8
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "wide_tiles" 2>&1 | grep -v "^$" | cut -c1-300 | tail -12 > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
for t in 0 18 19; do
  MMT_TILE_NARROW=$t timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-dense 2>$O/err_ab.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); r=[r for r in d['roofline_top3'] if 'down-proj' in r['kernel']][0]; print('narrow tile=$t  %.4f ms/step  %.0f pairs/s  FFN-down %.1f us frac %.3f' % (d['ms_per_step'], d['value'], r['avg_launch_us'], r['frac']))
"
done
done
for t in 0 19; do
  MMT_TILE_WIDE=$t timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-dense 2>$O/err_ab.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); r=[r for r in d['roofline_top3'] if 'up-proj' in r['kernel']][0]; print('wide tile=$t  %.4f ms/step  %.0f pairs/s  FFN-up %.1f us frac %.3f' % (d['ms_per_step'], d['value'], r['avg_launch_us'], r['frac']))
"
done
for t in 0 18; do
  MMT_TILE_NARROW=$t timeout 300 python bench.py --config 4 --steps 30 --warmup 5 --no-dense 2>$O/err_ab.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); print('config 4 narrow tile=$t  %.4f ms/step' % d['ms_per_step'], [(r['kernel'][:14], round(r['avg_launch_us'],1), round(r['frac'],3)) for r in d['roofline_top3']])
"
done
