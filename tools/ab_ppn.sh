# Same-box A/B of MMT_TILE_PPN (the persistent kernel on the long-K narrow GEMMs): headline + unpacked, configs[3], configs[4].
#   gpurun -- 'bash tools/ab_ppn.sh'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_ppn
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for v in 0 1; do
    MMT_TILE_PPN=$v timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>$O/err_$v.log | tail -1 > $O/bench_${v}_$rep.json
    MMT_TILE_PPN=$v timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline --no-dense 2>$O/err3_$v.log | tail -1 > $O/bench3_${v}_$rep.json
    python -c "
import json; d = json.load(open('$O/bench_${v}_$rep.json')); e = json.load(open('$O/bench3_${v}_$rep.json')); print('MMT_TILE_PPN=$v packed %.4f  unpacked %.4f  configs[3] %.4f ms/step' % (d['ms_per_step'], d['dense']['ms_per_step'], e['ms_per_step']))" | tee -a $O/summary.txt
  done
done
for v in 0 1; do
  MMT_TILE_PPN=$v timeout 600 python bench.py --config 4 --steps 30 --warmup 5 --no-cpu-baseline --no-dense 2>$O/err4_$v.log | tail -1 > $O/bench4_$v.json
  python -c "
import json; e = json.load(open('$O/bench4_$v.json')); print('MMT_TILE_PPN=$v configs[4] %.4f ms/step' % e['ms_per_step'])" | tee -a $O/summary.txt
done
