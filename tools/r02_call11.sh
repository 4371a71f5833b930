cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c11
mkdir -p $O
cd $R
for i in 1 2; do
MMT_TILE_192=0 timeout 600 python bench.py --steps 200 --warmup 15 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_off_$i.json
MMT_TILE_192=1 timeout 600 python bench.py --steps 200 --warmup 15 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_on_$i.json
done
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $O/pytest_gpu.txt
