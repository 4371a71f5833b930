cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r04_34
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "wgrad_grouped_256" > gpurun_out/r04_34/pytest.txt 2>&1
grep -n "^E  \|passed\|failed" gpurun_out/r04_34/pytest.txt | cut -c1-220 | head -30
