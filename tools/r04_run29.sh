cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "wide_tiles and 23" 2>&1 | tail -4 | cut -c1-200
timeout 300 python tools/gemm_lab.py --tiles 13,18,23 --rows 3573 --instep --nocheck 2>&1 | tail -9 | cut -c1-200
for t in 18 23 18 23; do MMT_TILE_NARROW=$t timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('narrow tile $t  %.4f ms/step' % d['ms_per_step'])"; done
