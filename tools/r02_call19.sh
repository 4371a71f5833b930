cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c19
mkdir -p $O
cd $R
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dense --force-collectives --grad-sync staged > $O/fc.txt 2>&1
echo "rc=$?" >> $O/fc.txt
