# gemm5 iteration: parity, lab (in-step launch form), whole-step A/B of MMT_TILE_PP
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wide_tiles and 24" 2>&1 | tail -3
timeout 300 python tools/gemm_lab.py --tiles 14,24 --rows 3639 --instep --nocheck 2>&1 | grep -v amdgpu.ids | head -5
for rep in 1 2; do
  for v in 0 1 2; do
    MMT_TILE_PP=$v timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>$O/err_$v.log | tail -1 > $O/bench_${v}_$rep.json
    python -c "
import json; d = json.load(open('$O/bench_${v}_$rep.json')); print('MMT_TILE_PP=$v %.4f ms/step  %.0f pairs/s  dense %.4f  loss %s' % (d['ms_per_step'], d['value'], d['dense']['ms_per_step'] if d.get('dense') else 0, d.get('first_loss')))"
  done
done
