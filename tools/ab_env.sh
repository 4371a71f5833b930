# Same-box A/B of one environment switch, alternating: bash tools/ab_env.sh VAR A B [bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_env
mkdir -p $O
cd $R
VAR=$1; A=$2; B=$3; shift 3
for rep in 1 2 3; do
  for v in $A $B; do
    env $VAR=$v timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline "$@" 2>$O/err_$v.log | tail -1 > $O/bench_${v}_$rep.json
    python -c "
import json; d = json.load(open('$O/bench_${v}_$rep.json')); print('$VAR=$v %.4f ms/step  %.0f pairs/s  dense %.4f  loss %s' % (d['ms_per_step'], d['value'], d['dense']['ms_per_step'] if d.get('dense') else 0, d.get('first_loss')))" | tee -a $O/summary.txt
  done
done
