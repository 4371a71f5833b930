cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_nn or splitk" 2>&1 | tail -12 | cut -c1-400
