# gpurun_out/r06/final/* (scratch, merged back by gpurun) -> profiles/r06_* (tracked)
set -e
cd "$(dirname "$0")/.."
for f in gpurun_out/r06/final/*.json gpurun_out/r06/final/*.txt gpurun_out/r06/final/*.csv; do
  [ -f "$f" ] || continue
  b=$(basename $f)
  case $b in
    prof_bench_*) ;;
    pmc_*|kernel_stats_*_by_grid.csv) cp $f profiles/r06_$b ;;
    fill_*) ;;
    bench_*) cp $f profiles/r06_final_$b ;;
    *) cp $f profiles/r06_final_$b ;;
  esac
done
python - <<'PY'
import glob, json, os
rows = {}
for p in sorted(glob.glob('gpurun_out/r06/final/fill_*.json')):
  name = os.path.basename(p)[5:-5]
  fill, policy = name.split('_', 1)
  try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    rows.setdefault(fill, {})[policy] = (d['ms_per_step'], d['config'].get('live_rows_rank0'))
  except Exception as e:
    rows.setdefault(fill, {})[policy] = (None, str(e)[:40])
with open('profiles/r06_fill_sweep.txt', 'w') as f:
  f.write('# bench.py --fill F --steps 200 --warmup 10 (same box, one after the other): the dispatcher with the loader\'s live-row counts\n'
          '# (auto) against three forced tile policies -- as_low_fill: every packed GEMM priced at 0.3 of its rows (narrow GEMMs on the\n'
          '# one-round phased tile 18, wide on tile 14); narrow13: the narrow GEMMs on the 8-wave tile at two blocks per CU; as_dense:\n'
          '# priced at all rows (tiles 24 / 13).  ms per step; "vs best" = auto / the best forced policy.\n')
  f.write('%-6s %-10s %10s %12s %10s %10s %8s\n' % ('fill', 'live rows', 'auto', 'as_low_fill', 'narrow13', 'as_dense', 'vs best'))
  for fill in sorted(rows, key=float):
    r = rows[fill]
    ms = {k: v[0] for k, v in r.items()}
    forced = [ms[k] for k in ('as_low_fill', 'narrow13', 'as_dense') if ms.get(k)]
    best = min(forced) if forced else None
    f.write('%-6s %-10s %10s %12s %10s %10s %8s\n' % (fill, r.get('auto', (None, None))[1],
            *['%.4f' % ms[k] if ms.get(k) else '-' for k in ('auto', 'as_low_fill', 'narrow13', 'as_dense')],
            '%.3f' % (ms['auto'] / best) if best and ms.get('auto') else '-'))
print(open('profiles/r06_fill_sweep.txt').read())
PY
ls profiles | grep r06 | wc -l
