cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "wide_tiles and 23" 2>&1 | tail -1 | cut -c1-200
for i in 1 2 3; do
for v in "18 0" "18 23"; do set -- $v
MMT_TILE_NARROW=$1 MMT_TILE_LONGK=$2 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('narrow $1 longk $2  %.4f ms/step' % d['ms_per_step'])"; done; done
