cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_1
mkdir -p $O
cd $R
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/gemm2_budget.py --alt > $O/gemm2_budget.txt 2>&1
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py > $O/attn_budget.txt 2>&1
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py --dense > $O/attn_budget_dense.txt 2>&1

cat $O/gemm2_budget.txt $O/attn_budget.txt $O/attn_budget_dense.txt

