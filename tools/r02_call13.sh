cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c13
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x  2>&1 | tail -40 > $O/pytest_gpu.txt
