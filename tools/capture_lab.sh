# 1-rank RCCL group: the multi-rank step with and without the collectives captured into the graph
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/capture
mkdir -p $O
cd $R
for v in "staged" "single"; do
timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-dense --force-collectives --grad-sync $v > $O/plain_$v.txt 2>&1; echo "rc=$?" >> $O/plain_$v.txt
timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-dense --force-collectives --grad-sync $v --capture-collectives > $O/captured_$v.txt 2>&1; echo "rc=$?" >> $O/captured_$v.txt
done
