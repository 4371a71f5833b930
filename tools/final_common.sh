# shared by tools/final_{1,2,3}.sh (sourced): where things go and the two profiling helpers
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
P=$R/profiles
mkdir -p $O
prof() {  # name, bench args: rocprofv3 kernel trace of the bench command -> by-grid summary (txt + csv), the replayed step's sequence
  rm -rf /tmp/prof && (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o step -- python $R/bench.py $2 --no-cpu-baseline --no-dense > $O/prof_$1.log 2>&1)
  DB=$(find /tmp/prof -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $DB --by-grid --top 90 --csv $O/kernel_stats_$1_by_grid.csv > $O/kernel_stats_$1_by_grid.txt 2>&1
  python $R/tools/graph_sequence.py $DB > $O/graph_sequence_$1.txt 2>&1
  cp $O/kernel_stats_$1_by_grid.csv $P/r05_kernel_stats_$1_by_grid.csv
}
pmc() {  # name, bench args: PMC counters, one pass per counter group (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass;
         # --pmc never together with a trace domain)
  rm -rf /tmp/pmc1 /tmp/pmc2 /tmp/pmc3
  (cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc1 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1)
  (cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc2 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1)
  (cd /tmp && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d /tmp/pmc3 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1)
  python $R/tools/rocpd_pmc.py $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*.db") --csv $O/pmc_$1.csv --top 40 > $O/pmc_$1.txt 2>&1
  cp $O/pmc_$1.csv $P/r05_pmc_$1.csv
}
line() {  # print one bench JSON file as a line
  python - $1 <<'PY'
import sys, json
try:
  d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print('%-44s %.4f ms/step  %.0f %s' % (sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d['unit']))
except Exception as e:
  print(sys.argv[1], 'UNREADABLE', e)
PY
}
