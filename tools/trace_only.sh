# Only the replayed-step kernel sequences of tools/final_profiles.sh (packed, config 3, config 4, fork 117).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
prof() {  # name, bench args
  rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o step -- python $R/bench.py $2 --no-cpu-baseline --no-dense > $O/prof_$1.log 2>&1
  DB=$(find /tmp/prof -name "*.db" | head -1)
  python $R/tools/graph_sequence.py $DB > $O/graph_sequence_$1.txt 2>&1
}
prof packed "--steps 50 --warmup 10"
prof config3 "--config 3 --steps 30 --warmup 5"
prof config4 "--config 4 --steps 15 --warmup 3"
prof fork117 "--steps 50 --warmup 10 --fork 117"
tail -2 $O/graph_sequence_*.txt
