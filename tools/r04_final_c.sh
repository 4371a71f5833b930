# bench lines again now that profiles/r04_pmc_*.csv exist (the `traffic` fields read them)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_final
mkdir -p $O
cd $R
timeout 900 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench_packed.json
timeout 600 python bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3.json
timeout 600 python bench.py --config 4 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config4.json
timeout 600 python bench.py --steps 200 --warmup 20 --dense --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_dense.json
for i in 1 2; do
timeout 600 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_resident_$i.json
timeout 600 python bench.py --steps 300 --warmup 20 --host-inputs --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_host_inputs_$i.json
timeout 600 python bench.py --steps 300 --warmup 20 --host-inputs --ragged-inputs --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_host_inputs_ragged_$i.json
done
for f in $O/bench_*.json; do python - $f <<'PY'
import sys, json
try:
  d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print('%-44s %.4f ms/step  %.0f %s' % (sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d['unit']))
except Exception as e:
  print(sys.argv[1], 'UNREADABLE', e)
PY
done
