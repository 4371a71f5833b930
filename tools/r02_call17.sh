cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c17
mkdir -p $O
cd $R
timeout 600 python tools/gemm_lab.py --rows 3596 --tiles 13,18,19 > $O/lab_plain.txt 2>&1
timeout 600 python tools/gemm_lab.py --rows 3596 --tiles 13,18,19 --instep --nocheck > $O/lab_instep.txt 2>&1
