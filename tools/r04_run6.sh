cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -2
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py 2>&1 | cut -c1-300 | head -3
bash tools/ab_prev.sh
