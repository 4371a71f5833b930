# tile sweeps on the bigger BASELINE shapes: bash tools/ab_cfg.sh CONFIG VAR "v1 v2 ..." [steps]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_cfg
mkdir -p $O
cd $R
CFG=$1; VAR=$2; VALS=$3; STEPS=${4:-30}
for v in $VALS; do
  if [ "$v" = "-" ]; then E=""; else E="$VAR=$v"; fi
  env $E timeout 300 python bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline --no-dense --input-slots 2 2>$O/err.log | tail -1 > $O/bench.json
  python -c "
import json; d = json.load(open('$O/bench.json')); print('config $CFG $VAR=$v %.4f ms/step  %.0f pairs/s  top3: %s' % (d['ms_per_step'], d['value'], ' | '.join('%.0f us %.2f' % (r['avg_launch_us'], r['frac']) for r in d['roofline_top3'])))" | tee -a $O/summary.txt
done
