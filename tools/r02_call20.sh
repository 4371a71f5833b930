cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c20
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $O/pytest_gpu.txt
timeout 600 python bench.py --steps 200 --warmup 15 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench.json
cd /tmp
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof -o step -- python $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-dense > $O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/tools/graph_sequence.py $DB > $O/graph_sequence.txt 2>&1
