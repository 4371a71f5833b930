cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_rowblock
mkdir -p $O
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o rb -- python $R/tools/large_sim_bench.py --iters 3 > $O/prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB --by-grid --top 40 > $O/kernel_stats_by_grid.txt 2>&1
tail -2 $O/prof.log
head -45 $O/kernel_stats_by_grid.txt | cut -c1-200
