# Round-5 evidence, part 3 (at HEAD): the GPU test suite, the smoke test, the default bench line.   gpurun -- 'bash tools/final_3.sh'
source "$(dirname "$0")/final_common.sh"
cd $R
(time timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $O/pytest_gpu_full.txt 2>&1; tail -22 $O/pytest_gpu_full.txt) > $O/pytest_gpu.txt 2>&1
grep -n "what()\|Error:\|error:\|terminated with\|HIP error\|NCCL" $O/pytest_gpu_full.txt | cut -c1-400 | head -30 > $O/pytest_gpu_errors.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -3 > $O/smoke.txt
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_default_head.json
cat $O/pytest_gpu_errors.txt $O/pytest_gpu.txt $O/smoke.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/final/bench_default_head.json').read().strip().splitlines()[-1])
print('default bench at HEAD: %.4f ms/step  %.0f pairs/s  steps %d  dense %s  cpu %.1f' % (d['ms_per_step'], d['value'], d['steps'], d['dense'], d['cpu_baseline']['value']))
r = d['roofline']; print(r['kernel'][:50], 'eager %.1f us frac %.3f | graph %s us frac %s | traffic %s vs %s' % (r['avg_launch_us'], r['frac'], r.get('avg_launch_us_graph'), r.get('frac_graph'), r['traffic'], r['algorithmic_bytes_per_launch']))
PY
