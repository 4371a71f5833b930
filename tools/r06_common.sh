# shared by tools/r06_*.sh (sourced): where things go and the profiling helpers.  Everything lands in gpurun_out/r06/ (scratch,
# merged back by gpurun); tools/collect_profiles.sh copies what is to be judged into profiles/r06_*.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
P=$R/profiles
mkdir -p $O
BENCH="--no-cpu-baseline --no-dense"
prof() {  # name, bench args [, env assignments]: rocprofv3 kernel trace of the bench command -> by-grid summary (txt + csv) + the replayed step's sequence
  rm -rf /tmp/prof && (cd /tmp && env $3 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o step -- python $R/bench.py $2 $BENCH > $O/prof_$1.log 2>&1)
  DB=$(find /tmp/prof -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $DB --by-grid --top 90 --csv $O/kernel_stats_$1_by_grid.csv > $O/kernel_stats_$1_by_grid.txt 2>&1
  python $R/tools/graph_sequence.py $DB > $O/graph_sequence_$1.txt 2>&1
  tail -1 $O/prof_$1.log > $O/prof_bench_$1.json
}
pmc1() {  # tag, counters (one pass), bench args [, env]: one rocprofv3 --pmc pass (never together with a trace domain)
  rm -rf /tmp/pmc_$1 && (cd /tmp && env $4 timeout 900 rocprofv3 --pmc $2 -d /tmp/pmc_$1 -o p -- python $R/bench.py $3 $BENCH > /dev/null 2>&1)
}
pmc_sum() {  # name, tags...: the passes' databases -> one csv + txt
  n=$1; shift
  dbs=""
  for t in "$@"; do dbs="$dbs $(find /tmp/pmc_$t -name '*.db' | head -1)"; done
  python $R/tools/rocpd_pmc.py $dbs --csv $O/pmc_$n.csv --top 45 > $O/pmc_$n.txt 2>&1
}
pmc() {  # name, bench args [, env]: the traffic / MFMA passes of r01-r05 (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass)
  pmc1 $1_a FETCH_SIZE "$2" "$3"
  pmc1 $1_b WRITE_SIZE "$2" "$3"
  pmc1 $1_c "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "$2" "$3"
  pmc_sum $1 $1_a $1_b $1_c
}
pmc_cache() {  # name, bench args [, env]: L2 hit / miss, fabric-side read and write requests with their outstanding-request integrals
  pmc1 $1_h "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "$2" "$3"
  pmc1 $1_r "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum" "$2" "$3"
  pmc1 $1_w "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum" "$2" "$3"
  pmc1 $1_s "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE" "$2" "$3"
  pmc_sum $1 $1_h $1_r $1_w $1_s
}
line() {  # print one bench JSON file as a line
  python - $1 <<'PY'
import sys, json
try:
  d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  r = d['config'].get('adam_riders')
  print('%-44s %.4f ms/step  %.0f %s  riders %s' % (sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d['unit'],
        ['%.2f' % x['ridden_fraction'] for x in r] if r else None))
except Exception as e:
  print(sys.argv[1], 'UNREADABLE', e)
PY
}
ab() {  # tag, bench args, env assignments: one bench line into $O/ab_<tag>.json
  (cd $R && env $3 timeout 600 python bench.py $2 $BENCH > $O/ab_$1.log 2>&1; tail -1 $O/ab_$1.log > $O/ab_$1.json; line $O/ab_$1.json)
}
