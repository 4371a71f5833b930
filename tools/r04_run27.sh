cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "eight_phase" 2>&1 | tail -2
timeout 300 python tools/gemm3_lab.py 2>&1 | tail -5 | cut -c1-260
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/gemm3_budget.py 2>&1 | sed -n 2,6p
