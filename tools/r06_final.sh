# Round 6 evidence: bench lines, rocprofv3 kernel traces and PMC passes for every bench shape, the fill sweep against forced
# tile policies, GPU test and smoke logs.  Everything lands in gpurun_out/r06/final_* ; tools/collect_profiles.sh copies it to
# profiles/r06_*.      gpurun --timeout 3000 -- 'bash tools/r06_final.sh'
source "$(dirname "$0")/r06_common.sh"
cd $R
mkdir -p $O/final
O=$O/final
fb() {  # name, bench args, env assignments: one bench line into $O/<name>.json
  (cd $R && env $3 timeout 900 python bench.py $2 $BENCH > $O/$1.log 2>&1; tail -1 $O/$1.log > $O/$1.json; line $O/$1.json)
}
# traces and counters FIRST: they land in profiles/r06_* on the box, so that the bench lines below quote this run's files
prof packed "--steps 50 --warmup 10"
prof dense "--dense --steps 40 --warmup 8"
prof config3 "--config 3 --steps 30 --warmup 5"
prof config4 "--config 4 --steps 15 --warmup 3"
prof native_text_tower "--text-tower native --steps 30 --warmup 5"
for n in packed dense config3 config4 native_text_tower; do cp $O/kernel_stats_${n}_by_grid.csv $P/r06_kernel_stats_${n}_by_grid.csv; done
pmc kernels "--steps 12 --warmup 3"; cp $O/pmc_kernels.csv $P/r06_pmc_kernels.csv
pmc dense "--dense --steps 10 --warmup 3"; cp $O/pmc_dense.csv $P/r06_pmc_dense.csv
pmc config3 "--config 3 --steps 6 --warmup 2"; cp $O/pmc_config3.csv $P/r06_pmc_config3.csv
pmc config4 "--config 4 --steps 4 --warmup 2"; cp $O/pmc_config4.csv $P/r06_pmc_config4.csv
pmc native_text_tower "--text-tower native --steps 6 --warmup 2"
BENCH="--no-dense"
(cd $R && timeout 900 python bench.py --steps 400 --warmup 20 > $O/bench_default_head.log 2>&1; tail -1 $O/bench_default_head.log > $O/bench_default_head.json; line $O/bench_default_head.json)
BENCH="--no-cpu-baseline --no-dense"
for i in 1 2; do fb bench_packed_$i "--steps 400 --warmup 20" "MMT_X=0"; done
fb bench_dense "--dense --steps 200 --warmup 20" "MMT_X=0"
fb bench_config3 "--config 3 --steps 100 --warmup 10" "MMT_X=0"
fb bench_config4 "--config 4 --steps 30 --warmup 5" "MMT_X=0"
base=$(python -c "import json; print(json.load(open('$O/bench_packed_1.json'))['ms_per_step'])")
fb bench_native_text_tower "--text-tower native --steps 100 --warmup 10 --tower-base-ms $base" "MMT_X=0"
fb bench_native_text_tower_r05_launches "--text-tower native --steps 100 --warmup 10" "MMT_SPLITK_LN=0"
fb bench_host_inputs "--host-inputs --steps 200 --warmup 20" "MMT_X=0"
fb bench_host_inputs_ragged "--host-inputs --ragged-inputs --steps 200 --warmup 20" "MMT_X=0"
fb bench_adam_riders "--adam-riders --steps 200 --warmup 20" "MMT_X=0"
(cd $R && MMT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-dense --grad-algo rs_ag --shard-optimizer > $O/bench_2ranks_gloo.log 2>&1; tail -1 $O/bench_2ranks_gloo.log > $O/bench_2ranks_gloo_sharded_adam.json; line $O/bench_2ranks_gloo_sharded_adam.json)
# fill sweep: the dispatcher with the loader's live-row counts against three forced policies (narrow GEMMs always one-round phased /
# always two blocks per CU / everything priced as dense rows)
for f in 0.25 0.51 0.75 1.0; do
  fb fill_${f}_auto "--fill $f --steps 200 --warmup 10" "MMT_X=0"
  fb fill_${f}_as_low_fill "--fill $f --steps 200 --warmup 10" "MMT_LIVE_FRACTION=0.3"
  fb fill_${f}_narrow13 "--fill $f --steps 200 --warmup 10" "MMT_TILE_NARROW=13 MMT_TILE_PPN=0"
  fb fill_${f}_as_dense "--fill $f --steps 200 --warmup 10" "MMT_LIVE_FRACTION=1.0"
done
(cd $R && timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt)
(cd $R && timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt)
ls $O | wc -l
