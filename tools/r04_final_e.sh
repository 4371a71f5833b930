# last check at HEAD: GPU suite, smoke, the default bench line
bash tools/r04_final_a.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r04_final/bench_default_head.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_final/bench_default_head.json').read().strip().splitlines()[-1])
print('default bench at HEAD: %.4f ms/step  %.0f pairs/s  steps %d  dense %s  cpu %.1f' % (d['ms_per_step'], d['value'], d['steps'], d['dense'], d['cpu_baseline']['value']))
PY
