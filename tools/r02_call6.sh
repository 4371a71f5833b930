cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c6
mkdir -p $O
cd $R
TOPN=14 timeout 900 python tools/grad_parity_lab.py tiny configA configB > $O/grad_parity.txt 2>&1
