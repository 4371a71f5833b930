cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm2_tiles or gemm_nt_tiles or tile" 2>&1 | tail -2
timeout 300 python tools/gemm_lab.py --tiles 13,18,22 --rows 3573 --instep --nocheck 2>&1 | tail -9 | cut -c1-200
for t in 18 22; do MMT_TILE_NARROW=$t timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('narrow tile $t  %.4f ms/step' % d['ms_per_step'])"; done
for t in 18 22; do MMT_TILE_NARROW=$t timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('narrow tile $t  %.4f ms/step' % d['ms_per_step'])"; done
