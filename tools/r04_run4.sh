cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_4
mkdir -p $O
cd $R
for i in 1 2; do
for sp in 0 2; do
MMT_ATTN_BWD_SPLIT=$sp timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('split=$sp', d['ms_per_step'], d['value'])"
done
done
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so MMT_ATTN_BWD_SPLIT=2 timeout 300 python tools/attn_budget.py 2>&1 | cut -c1-400 | head -12
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py --fwd 2>&1 | cut -c1-1200 | head -12
