# gemm5 (tile 24): parity, lab timings against tile 14 / 18, cycle budget
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wide_tiles and 24" 2>&1 | tail -5 | tee $O/pytest.txt
timeout 300 python tools/gemm_lab.py --tiles 14,18,24 --rows 3639 --instep 2>&1 | grep -v amdgpu.ids | tee $O/lab_instep.txt
MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/g5_budget.py 3639 2>&1 | grep -v amdgpu.ids | tee $O/budget_warm.txt
