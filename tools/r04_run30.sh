cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "wide_tiles and 23" 2>&1 | tail -2 | cut -c1-200
timeout 300 python tools/gemm_lab.py --tiles 13,18,23 --rows 3573 --instep --nocheck 2>&1 | tail -9 | cut -c1-200
