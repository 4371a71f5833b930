# Round artifacts (r03): bench JSON lines (configs 1 / 3 / 4, dense, native text tower, host inputs, 2 gloo ranks), rocprofv3
# kernel-trace summaries of the same bench commands, the replayed step's kernel sequence, PMC passes (separate runs: --pmc
# never together with a trace domain), the fork-mode A/B, GPU test + smoke logs.
# Run on the GPU box:  gpurun -- 'bash tools/final_profiles.sh';  then  bash tools/collect_profiles.sh  copies the
# summaries into profiles/r03_*.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.txt 2>&1
timeout 900 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench_packed.json
timeout 600 python bench.py --steps 200 --warmup 20 --dense --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_dense.json
timeout 600 python bench.py --config 3 --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bench_config3.json
timeout 600 python bench.py --config 4 --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_config4.json
timeout 600 python bench.py --steps 100 --warmup 10 --text-tower native --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_native_text_tower.json
timeout 600 python bench.py --steps 200 --warmup 20 --host-inputs --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_host_inputs.json
timeout 600 python bench.py --steps 200 --warmup 20 --host-inputs --ragged-inputs --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_host_inputs_ragged.json
timeout 600 python bench.py --steps 200 --warmup 20 --input-slots 1 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_one_input_slot.json
MMT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_2ranks_gloo_one_gpu.json
for f in 0 32 16 1 21 117; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense --fork $f 2>/dev/null | tail -1 | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); print('fork %3s  %.4f ms/step  %.0f pairs/s' % ('$f', d['ms_per_step'], d['value']))
" >> $O/fork_lab.txt
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
prof fork117 "--steps 50 --warmup 10 --fork 117"
# PMC: one pass per counter group (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass)
pmc() {  # name, bench args
  rm -rf /tmp/pmc1 /tmp/pmc2 /tmp/pmc3
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc1 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc2 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d /tmp/pmc3 -o p -- python $R/bench.py $2 --no-cpu-baseline --no-dense > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*.db") --csv $O/pmc_$1.csv --top 40 > $O/pmc_$1.txt 2>&1
}
pmc kernels "--steps 12 --warmup 3"
pmc config3 "--config 3 --steps 6 --warmup 2"
pmc config4 "--config 4 --steps 4 --warmup 2"
ls -la $O
