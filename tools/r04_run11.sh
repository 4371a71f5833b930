cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do
for big in 0 1; do
MMT_TILE_BIG=$big timeout 600 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); rb=d.get('similarity_loss_row_block',{}); print('big=$big config4 %.3f ms/step  %.0f pairs/s | row block %.2f ms mfma_frac %.3f | exec frac %s' % (d['ms_per_step'], d['value'], rb.get('ms_fwd_bwd',0), rb.get('mfma_frac',0), d.get('executed_mfma_frac')))"
done
done
MMT_TILE_BIG=0 timeout 600 python bench.py --config 3 --steps 40 --warmup 5 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('big=0 config3', d['ms_per_step'])"
MMT_TILE_BIG=1 timeout 600 python bench.py --config 3 --steps 40 --warmup 5 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('big=1 config3', d['ms_per_step'])"
timeout 900 python -m pytest tests/test_cenet_gpu.py tests/test_large_sim_gpu.py -x -q 2>&1 | tail -3
