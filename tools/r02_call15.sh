cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c15
mkdir -p $O
cd $R
for reg in 0 1; do
MMT_WGRAD_REG=$reg MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/wgrad_instr.py > $O/wgrad_instr_$reg.txt 2>&1
MMT_WGRAD_REG=$reg timeout 300 python tools/wgrad_instr.py > $O/wgrad_plain_$reg.txt 2>&1
done
MMT_WGRAD_REG=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "wgrad or grad" 2>&1 | tail -4 > $O/pytest_reg.txt
