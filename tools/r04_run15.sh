cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== shipped kernels"; TOPN=40 timeout 600 python tools/grad_parity_lab.py configB 2>&1 | grep -E "pack=|query.weight|key.weight|value.weight|median" | head -24
echo "== dS kept as hi + lo bf16 through the dQ / dK products (lab build)"; MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so TOPN=40 timeout 600 python tools/grad_parity_lab.py configB 2>&1 | grep -E "pack=|query.weight|key.weight|value.weight|median" | head -24
