cd /root/repo
timeout 900 python -m pytest tests/test_cenet_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -5 | cut -c1-400
