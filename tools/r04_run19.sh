cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_large_sim_gpu.py tests/test_kernels_gpu.py tests/test_cenet_gpu.py -x -q -k "large or row_sharded or transpose or sim" 2>&1 | tail -3
bash tools/r04_run18.sh
