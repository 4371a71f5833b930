# kernel-trace summary + replayed step sequence of the default bench -> gpurun_out/$1/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04_prof}
mkdir -p $O
rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o step -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-dense ${2:-} > $O/prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB --by-grid --top 90 > $O/kernel_stats_by_grid.txt 2>&1
python $R/tools/graph_sequence.py $DB > $O/graph_sequence.txt 2>&1
tail -1 $O/prof.log | cut -c1-300
head -50 $O/kernel_stats_by_grid.txt
