# Same-box A/B of two bench command lines, alternating: bash tools/ab_args.sh "<args A>" "<args B>"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_args
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for w in A B; do
    if [ $w = A ]; then ARGS="$1"; else ARGS="$2"; fi
    timeout 600 python bench.py --steps 300 --warmup 20 --no-cpu-baseline $ARGS 2>$O/err_$w.log | tail -1 > $O/bench_${w}_$rep.json
    python -c "
import json; d = json.load(open('$O/bench_${w}_$rep.json')); print('[$ARGS] %.4f ms/step  %.0f pairs/s  dense %.4f' % (d['ms_per_step'], d['value'], d['dense']['ms_per_step']))" | tee -a $O/summary.txt
  done
done
