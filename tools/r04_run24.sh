cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r04_24
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > gpurun_out/r04_24/pytest.txt 2>&1
tail -3 gpurun_out/r04_24/pytest.txt
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py --nosched 2>&1 | cut -c1-330 | tail -13
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py 2>&1 | cut -c1-330 | tail -13
