# A/B of two builds of the library on one box, alternating: mmt_amd/lib/libmmt_hip_base.so (HEAD) against the tree's.
# gpurun -- 'bash tools/ab_lib.sh [extra bench args]'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_lib
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_base.so; else unset MMT_HIP_LIB; fi
    timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline "$@" 2>$O/err_$which.log | tail -1 > $O/bench_${which}_$rep.json
    python -c "
import json; d = json.load(open('$O/bench_${which}_$rep.json')); print('$which %.4f ms/step  %.0f pairs/s  dense %.4f' % (d['ms_per_step'], d['value'], d['dense']['ms_per_step']))" | tee -a $O/summary.txt
  done
done
