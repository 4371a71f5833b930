cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c9
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ragged_gpu.py -q -m gpu -x 2>&1 | tail -25 > $O/pytest_gpu.txt
timeout 600 python tools/feed_lab.py > $O/feed_lab.txt 2>&1
timeout 600 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-dense --host-inputs 2>$O/bench_host.err | tail -1 > $O/bench_host.json
timeout 600 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --ragged-inputs --host-inputs 2>$O/bench_host_ragged.err | tail -1 > $O/bench_host_ragged.json
