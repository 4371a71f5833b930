# Round-5 evidence, part 1: rocprofv3 kernel traces and PMC passes for every bench shape (they land in profiles/r05_* ON THE BOX,
# so that part 2's bench lines quote THIS run's files), then the s_memtime budgets.   gpurun -- 'bash tools/final_1.sh'
source "$(dirname "$0")/final_common.sh"
cd $R
prof packed "--steps 50 --warmup 10"
prof dense "--dense --steps 40 --warmup 8"
prof config3 "--config 3 --steps 30 --warmup 5"
prof config4 "--config 4 --steps 15 --warmup 3"
pmc kernels "--steps 12 --warmup 3"
pmc dense "--dense --steps 10 --warmup 3"
pmc config3 "--config 3 --steps 6 --warmup 2"
pmc config4 "--config 4 --steps 4 --warmup 2"
cd $R
MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/gemm2_budget.py > $O/gemm2_budget.txt 2>&1
MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/g5_budget.py 3639 2>&1 | grep -v amdgpu.ids > $O/g5_budget_warm.txt
MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/g5_budget.py 3639 --cold 2>&1 | grep -v amdgpu.ids > $O/g5_budget_cold.txt
MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/g5_budget.py 6976 2>&1 | grep -v amdgpu.ids > $O/g5_budget_dense_rows.txt
tail -2 $O/graph_sequence_packed.txt; head -3 $O/pmc_kernels.txt; ls $O
