cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_cenet_gpu.py -x -q -k "attention or bench_configuration or packed_equals_dense" 2>&1 | grep -v "^$" | tail -8
ARGS="--steps 200 --warmup 20 --no-cpu-baseline --no-dense"
for i in 1 2 3; do
  for mode in 0 1; do
    MMT_ATTN_SCHED=$mode timeout 300 python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('sched %-3s %.4f ms/step  %.0f pairs/s' % ('$mode', d['ms_per_step'], d['value']))"
  done
done
MMT_ATTN_SCHED=0 timeout 300 python bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | cut -c1-200
MMT_ATTN_SCHED=1 timeout 300 python bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | cut -c1-200
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py 2>&1 | cut -c1-250 | tail -14
