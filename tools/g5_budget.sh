# gemm5 cycle budgets, operands warm and cold (768 MB fill in front of the measured launch)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g5
mkdir -p $O
cd $R
MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/g5_budget.py 3639 "$@" 2>&1 | grep -v amdgpu.ids | tee $O/budget_$1.txt
