cd /root/repo
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -30 | cut -c1-600
