# Same-box alternations of MMT_TILE_PPN=1 (default) against 6 (tile 25 on the packed N = 512, K < 1536 GEMMs).  gpurun -- 'bash tools/ab_ppn3.sh'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_ppn3
mkdir -p $O
cd $R
for rep in 1 2 3 4 5; do
  for v in 1 6; do
    MMT_TILE_PPN=$v timeout 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-dense 2>$O/err.log | tail -1 > $O/b.json
    python -c "
import json; d = json.load(open('$O/b.json')); print('MMT_TILE_PPN=$v packed %.4f ms/step' % d['ms_per_step'])" | tee -a $O/summary.txt
  done
done
