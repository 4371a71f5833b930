cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_12
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
