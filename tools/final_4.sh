# Round-5 evidence, part 4 (after the long-K narrow GEMMs moved to gemm5 on unpacked rows): traces + PMC of the shapes whose
# kernels changed (unpacked headline step, configs[3]), then their bench lines and the headline's.   gpurun -- 'bash tools/final_4.sh'
source "$(dirname "$0")/final_common.sh"
cd $R
prof dense "--dense --steps 40 --warmup 8"
prof config3 "--config 3 --steps 30 --warmup 5"
pmc dense "--dense --steps 10 --warmup 3"
pmc config3 "--config 3 --steps 6 --warmup 2"
cd $R
timeout 900 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench_packed.json
timeout 600 python bench.py --steps 200 --warmup 20 --dense --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_dense.json
timeout 600 python bench.py --config 3 --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bench_config3.json
for f in $O/bench_packed.json $O/bench_dense.json $O/bench_config3.json; do line $f; done
MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/g5_budget.py 6976 2>&1 | grep -v amdgpu.ids > $O/g5_budget_dense_rows.txt
head -12 $O/kernel_stats_dense_by_grid.txt
