cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c6
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $O/pytest_gpu.txt
