# Round artifacts (r02): bench JSON lines, rocprofv3 kernel-trace summary of the same bench command, the replayed step's
# kernel sequence, PMC passes (separate runs: --pmc never together with a trace domain), GPU test + smoke logs.
# Run on the GPU box:  gpurun -- 'bash tools/final_profiles.sh';  then  bash tools/collect_profiles.sh  copies the
# summaries into profiles/r02_final_*.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -2 > $O/pytest_gpu.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.txt 2>&1
timeout 900 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench_packed.json
timeout 600 python bench.py --steps 200 --warmup 20 --dense --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_dense.json
timeout 600 python bench.py --steps 100 --warmup 10 --text-tower native --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_native_text_tower.json
timeout 600 python bench.py --steps 200 --warmup 20 --host-inputs --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_host_inputs.json
timeout 600 python bench.py --steps 200 --warmup 20 --host-inputs --ragged-inputs --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_host_inputs_ragged.json
timeout 600 python bench.py --steps 200 --warmup 20 --ragged-inputs --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ragged.json
timeout 600 python tools/shape_bench.py > $O/other_shapes.txt 2>&1
cd /tmp
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o step -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-dense > $O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB --csv $O/kernel_stats_packed.csv --top 70 > $O/kernel_stats_packed.txt 2>&1
python $R/tools/rocpd_stats.py $DB --by-grid --top 90 > $O/kernel_stats_packed_by_grid.txt 2>&1
python $R/tools/graph_sequence.py $DB > $O/graph_sequence_packed.txt 2>&1
rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o step -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dense --text-tower native > $O/prof_bench_native.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB --by-grid --top 90 > $O/kernel_stats_native_text_tower_by_grid.txt 2>&1
# PMC: one pass per counter group (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass)
rm -rf /tmp/pmc1 /tmp/pmc2 /tmp/pmc3
timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc1 -o p -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-dense > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc2 -o p -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-dense > /dev/null 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d /tmp/pmc3 -o p -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-dense > /dev/null 2>&1
python $R/tools/rocpd_pmc.py $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*.db") --csv $O/pmc_kernels.csv --top 40 > $O/pmc_kernels.txt 2>&1
ls -la $O
