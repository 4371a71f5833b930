cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -3
timeout 300 python tools/gemm_lab.py --tiles 14,15,20,16 --rows 3639 --instep --nocheck 2>&1 | head -4
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/gemm2_budget.py 2>&1 | sed -n 2,7p | cut -c1-420
bash tools/ab_prev.sh
