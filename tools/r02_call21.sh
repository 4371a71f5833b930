cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c21
mkdir -p $O
cd $R
MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/wgrad_instr.py > $O/wgrad_instr.txt 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_cenet_gpu.py -q -m gpu -x 2>&1 | tail -4 > $O/pytest_gpu.txt
timeout 600 python bench.py --steps 200 --warmup 15 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench.json
