cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c10
mkdir -p $O
cd $R
MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/wgrad_instr.py > $O/wgrad_instr.txt 2>&1
