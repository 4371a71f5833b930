cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "eight_phase or gemm_nt_wide" 2>&1 | tail -12
timeout 600 python tools/gemm3_lab.py 2>&1 | tail -8
