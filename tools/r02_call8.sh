cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c8
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ragged_gpu.py tests/test_dp_gpu.py -q -m gpu -x 2>&1 | tail -25 > $O/pytest_gpu.txt
timeout 600 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-dense 2>$O/bench.err | tail -1 > $O/bench.json
timeout 600 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --ragged-inputs 2>$O/bench_ragged.err | tail -1 > $O/bench_ragged.json
timeout 600 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-dense --host-inputs 2>$O/bench_host.err | tail -1 > $O/bench_host.json
timeout 600 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --ragged-inputs --host-inputs 2>$O/bench_host_ragged.err | tail -1 > $O/bench_host_ragged.json
cd /tmp
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof -o step -- python $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-dense > $O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/tools/graph_sequence.py $DB > $O/graph_sequence.txt 2>&1
ls $O
