# Runtime switches that might change the per-node cost of a replayed HIP graph: one bench run per setting (same box).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/env_lab
mkdir -p $O
cd $R
run() {
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense --input-slots 4 2>$O/err.log | tail -1 > $O/bench.json
  python -c "
import json
try:
  d = json.load(open('$O/bench.json')); print('%-45s %.4f ms/step' % ('$*', d['ms_per_step']))
except Exception as e:
  print('%-45s FAILED' % '$*')" | tee -a $O/summary.txt
}
run X=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run ROC_SYSTEM_SCOPE_SIGNAL=0
run GPU_MAX_HW_QUEUES=1
run HSA_ENABLE_INTERRUPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=1000
run X=0
