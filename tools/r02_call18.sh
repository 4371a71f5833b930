cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c18
mkdir -p $O
cd $R
timeout 600 python tools/cold_lab.py > $O/cold_lab.txt 2>&1
