# The replayed step's kernel sequence under two settings of one environment switch (rocprofv3 kernel trace, same box):
#   bash tools/seq_ab.sh VAR A B [bench args]     -> gpurun_out/seq_ab/graph_sequence_{A,B}.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/seq_ab
mkdir -p $O
VAR=$1; A=$2; B=$3; shift 3
for v in $A $B; do
  rm -rf /tmp/prof_$v
  env $VAR=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o step -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-dense "$@" > $O/prof_$v.log 2>&1
  DB=$(find /tmp/prof_$v -name "*.db" | head -1)
  python $R/tools/graph_sequence.py $DB > $O/graph_sequence_$v.txt 2>&1
  python $R/tools/rocpd_stats.py $DB --by-grid --top 60 > $O/kernel_stats_${v}_by_grid.txt 2>&1
  tail -1 $O/graph_sequence_$v.txt
done
