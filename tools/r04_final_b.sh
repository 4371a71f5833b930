# Round-4 artifacts, part B: bench JSON lines (configs 1 / 3 / 4, native text tower, host inputs, 2 gloo ranks with the sharded
# optimizer), rocprofv3 kernel-trace summaries of the same bench commands, the replayed step's kernel sequence, PMC passes
# (separate runs: --pmc never together with a trace domain), the s_memtime budgets.  -> gpurun_out/r04_final/;
# tools/collect_profiles.sh copies the summaries into profiles/r04_*.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_final
mkdir -p $O
cd $R
timeout 900 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench_packed.json
timeout 600 python bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3.json
timeout 600 python bench.py --config 4 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config4.json
timeout 600 python bench.py --steps 100 --warmup 10 --text-tower native --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_native_text_tower.json
timeout 600 python bench.py --steps 200 --warmup 20 --host-inputs --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_host_inputs.json
timeout 600 python bench.py --steps 200 --warmup 20 --host-inputs --ragged-inputs --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_host_inputs_ragged.json
MMT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --grad-algo rs_ag --shard-optimizer 2>/dev/null | tail -1 > $O/bench_2ranks_gloo_sharded_adam.json
for f in $O/bench_*.json; do python - $f <<'PY'
import sys, json
try:
  d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print('%-44s %.4f ms/step  %.0f %s' % (sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d['unit']))
except Exception as e:
  print(sys.argv[1], 'UNREADABLE', e)
PY
done
cd /tmp
prof() {  # name, bench args
  rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o step -- python $R/bench.py $2 --no-cpu-baseline --no-dense > $O/prof_$1.log 2>&1
  DB=$(find /tmp/prof -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $DB --by-grid --top 90 > $O/kernel_stats_$1_by_grid.txt 2>&1
  python $R/tools/graph_sequence.py $DB > $O/graph_sequence_$1.txt 2>&1
}
prof packed "--steps 50 --warmup 10"
python $R/tools/rocpd_stats.py $(find /tmp/prof -name "*.db" | head -1) --csv $O/kernel_stats_packed.csv --top 70 > $O/kernel_stats_packed.txt 2>&1
prof config3 "--config 3 --steps 30 --warmup 5"
prof config4 "--config 4 --steps 15 --warmup 3"
# PMC: one pass per counter group (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass)
pmc() {  # name, bench args
  rm -rf /tmp/pmc1 /tmp/pmc2 /tmp/pmc3
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc1 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc2 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d /tmp/pmc3 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*.db") --csv $O/pmc_$1.csv --top 40 > $O/pmc_$1.txt 2>&1
}
pmc kernels "--steps 12 --warmup 3"
pmc config4 "--config 4 --steps 4 --warmup 2"
pmc config3 "--config 3 --steps 6 --warmup 2"
cd $R
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/gemm2_budget.py > $O/gemm2_budget.txt 2>&1
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py > $O/attn_budget.txt 2>&1
ls -la $O
