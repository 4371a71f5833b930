cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_2
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or 192" 2>&1 | tail -5 > $O/pytest_attn.txt
timeout 200 python tools/gemm_lab.py --tiles 20 --rows 3639 --instep > $O/gemm_lab_tile20.txt 2>&1
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/gemm2_budget.py > $O/gemm2_budget.txt 2>&1
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py > $O/attn_budget.txt 2>&1
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench.json
MMT_TILE_192=2 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_no192.json
cat $O/pytest_attn.txt $O/gemm_lab_tile20.txt $O/gemm2_budget.txt; cut -c1-1500 $O/attn_budget.txt
python - <<PY
import json
for f in ('bench','bench_no192'):
  d=json.load(open('$O/%s.json'%f)); print(f, d['ms_per_step'], d['value'])
PY
