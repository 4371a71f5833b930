cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_trace
mkdir -p $O
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o step -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-dense > $O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB --by-grid --top 70 > $O/kernel_stats_by_grid.txt 2>&1
python $R/tools/graph_sequence.py $DB > $O/graph_sequence.txt 2>&1
tail -1 $O/graph_sequence.txt
