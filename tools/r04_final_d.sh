# configs[4] artifacts again after the wgrad3 tile-order change: PMC passes, kernel summary, bench line
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_final
mkdir -p $O
prof() {
  rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o step -- python $R/bench.py $2 --no-cpu-baseline --no-dense > $O/prof_$1.log 2>&1
  DB=$(find /tmp/prof -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $DB --by-grid --top 90 > $O/kernel_stats_$1_by_grid.txt 2>&1
  python $R/tools/graph_sequence.py $DB > $O/graph_sequence_$1.txt 2>&1
}
pmc() {
  rm -rf /tmp/pmc1 /tmp/pmc2 /tmp/pmc3
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc1 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc2 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d /tmp/pmc3 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*.db") --csv $O/pmc_$1.csv --top 40 > $O/pmc_$1.txt 2>&1
}
prof config4 "--config 4 --steps 15 --warmup 3"
pmc config4 "--config 4 --steps 4 --warmup 2"
cd $R
cp $O/pmc_config4.csv profiles/r04_pmc_config4.csv
timeout 600 python bench.py --config 4 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config4.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_final/bench_config4.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['executed_mfma_frac'])
for r in d['roofline_top3']: print(r['kernel'][:40], round(r['avg_launch_us'], 1), round(r['frac'], 3), r['traffic'], r['algorithmic_bytes_per_launch'])
print(d['similarity_loss_row_block'])
PY
