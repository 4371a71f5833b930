# r03: full GPU suite + bench on the current tree.  gpurun -- 'bash tools/r03_check.sh'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_check
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
for rep in 1 2; do
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>$O/err.log | tail -1 > $O/bench_$rep.json
  python -c "
import json; d = json.load(open('$O/bench_$rep.json')); print('%.4f ms/step  %.0f pairs/s  dense %s' % (d['ms_per_step'], d['value'], d['dense']))"
done
