cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_cenet_gpu.py tests/test_text_bert_gpu.py -x -q -k "attention or cenet or text_bert or bench or packed" 2>&1 | tail -6
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py 2>&1 | cut -c1-420 | head -11
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/attn_budget.py --fwd 2>&1 | cut -c1-420 | sed -n 2,8p
bash tools/ab_prev.sh
