# bench.py --fill {0.25, 0.5, 0.75, 1.0} -> gpurun_out/r04_fill/fill_sweep.json (one bench line per fill)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_fill
mkdir -p $O
cd $R
: > $O/fill_sweep.json
for f in 0.25 0.5 0.75 1.0; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --fill $f 2>/dev/null | tail -1 >> $O/fill_sweep.json
done
python - <<PY
import json
for l in open('$O/fill_sweep.json'):
  d = json.loads(l)
  print('fill arg %.2f: token fill %.3f  %.4f ms/step  %.0f pairs/s  (dense step %.4f ms)  executed MFMA frac %.3f' % (
      d['config']['fill_arg'], d['fill_fraction'], d['ms_per_step'], d['value'], d['dense']['ms_per_step'], d['executed_mfma_frac']))
PY
