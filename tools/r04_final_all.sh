# everything of tools/r04_final_{a,b,c}.sh in one call (HEAD)
bash tools/r04_final_a.sh
bash tools/r04_final_b.sh > /dev/null 2>&1
cd ${GRAFT_REPO_ROOT:-/root/repo}
# the PMC files the bench lines' `traffic` fields read are those of THIS run
cp gpurun_out/r04_final/pmc_kernels.csv profiles/r04_pmc_kernels.csv
cp gpurun_out/r04_final/pmc_config3.csv profiles/r04_pmc_config3.csv
cp gpurun_out/r04_final/pmc_config4.csv profiles/r04_pmc_config4.csv
bash tools/r04_final_c.sh
