# Same-box comparison of several values of one environment switch: bash tools/ab_multi.sh VAR "v1 v2 v3" [reps] [bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_multi
mkdir -p $O
cd $R
VAR=$1; VALS=$2; REPS=${3:-2}; shift 3
for rep in $(seq 1 $REPS); do
  for v in $VALS; do
    if [ "$v" = "-" ]; then E=""; else E="$VAR=$v"; fi
    env $E timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense --input-slots 4 "$@" 2>$O/err.log | tail -1 > $O/bench.json
    python -c "
import json; d = json.load(open('$O/bench.json')); print('$VAR=$v %.4f ms/step  %.0f pairs/s' % (d['ms_per_step'], d['value']))" | tee -a $O/summary.txt
  done
done
