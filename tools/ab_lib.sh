# Same-box A/B of two builds of the library, alternating: bash tools/ab_lib.sh <libA.so> <libB.so> [bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_lib
mkdir -p $O
cd $R
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for v in $A $B; do
    n=$(basename $v .so)
    MMT_HIP_LIB=$R/$v timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-dense "$@" 2>$O/err_$n.log | tail -1 > $O/bench_${n}_$rep.json
    python -c "
import json; d = json.load(open('$O/bench_${n}_$rep.json')); r = d['roofline']; print('$n %.4f ms/step  %.0f pairs/s  loss %s | %s %.1f us' % (d['ms_per_step'], d['value'], d.get('first_loss'), r['kernel'][:30], r['avg_launch_us']))" | tee -a $O/summary.txt
  done
done
