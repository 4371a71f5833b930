cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
MMT_SPLITK_FFN=21 timeout 600 python -m pytest tests/test_cenet_gpu.py -x -q -k "bench_configuration or packed_equals_dense or every_parameter" 2>&1 | tail -3
ARGS="--steps 200 --warmup 20 --no-cpu-baseline --no-dense"
for i in 1 2 3; do
  for mode in 0 21 20 42; do
    MMT_SPLITK_FFN=$mode timeout 300 python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('mode %-3s %.4f ms/step  %.0f pairs/s' % ('$mode', d['ms_per_step'], d['value']))"
  done
done
