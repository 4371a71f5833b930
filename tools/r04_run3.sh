cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_3
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -15 > $O/pytest_attn.txt
cat $O/pytest_attn.txt
timeout 900 python -m pytest tests/test_cenet_gpu.py tests/test_text_bert_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -15 > $O/pytest_more.txt
cat $O/pytest_more.txt
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py > $O/attn_budget.txt 2>&1
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py --fwd > $O/attn_budget_fwd.txt 2>&1
cut -c1-1600 $O/attn_budget.txt | head -14;  cut -c1-1600 $O/attn_budget_fwd.txt | head -8
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench.json
python - <<PY
import json
d=json.load(open('$O/bench.json')); print('bench', d['ms_per_step'], d['value'])
PY
