cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_large_sim_gpu.py tests/test_kernels_gpu.py tests/test_dp_gpu.py -x -q -k "large or table or sharded or Sharded or sim" 2>&1 | tail -4
timeout 600 python bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_config3.json
timeout 600 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_config4.json
python - <<PY
import json
for c in (3,4):
  d=json.load(open('$O/bench_config%d.json'%c)); print('config',c, d['ms_per_step'], d['value'], {k:v for k,v in d.items() if 'row_block' in k or 'similarity' in k})
PY
