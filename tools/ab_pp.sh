# whole-step A/B of the persistent wide-GEMM tile policy: MMT_TILE_PP = 0 / 1 (/ 2), alternating, same box; headline + configs[3]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_pp
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "wide_tiles and 24" 2>&1 | tail -2
for rep in 1 2; do
  for v in 0 1; do
    MMT_TILE_PP=$v timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>$O/err_$v.log | tail -1 > $O/bench_${v}_$rep.json
    python -c "
import json; d = json.load(open('$O/bench_${v}_$rep.json')); print('headline MMT_TILE_PP=$v %.4f ms/step  dense %.4f' % (d['ms_per_step'], d['dense']['ms_per_step'] if d.get('dense') else 0))"
    MMT_TILE_PP=$v timeout 300 python bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline --no-dense 2>>$O/err_$v.log | tail -1 > $O/bench3_${v}_$rep.json
    python -c "
import json; d = json.load(open('$O/bench3_${v}_$rep.json')); print('configs[3] MMT_TILE_PP=$v %.4f ms/step' % d['ms_per_step'])"
  done
done
