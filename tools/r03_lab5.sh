cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_lab5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_eval_loop_gpu.py tests/test_dp_gpu.py -q -m gpu -s -x -k "eval_loop or separated or forked or staged" 2>&1 | grep -v "^$" | cut -c1-300 | tail -30 > $O/pytest_eval.txt
cat $O/pytest_eval.txt | tail -30
timeout 600 python tools/shape_bench.py 2>&1 | tail -4
