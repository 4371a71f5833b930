# round 2, GPU call 1: tests of the new tiles / fused Adam / ln_bwd, tile lab, bench A/B (192-wide tiles), kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/pytest_gpu.txt
timeout 600 python tools/gemm_lab.py --rows 3596,6976 --tiles 13,14,15,16,17 --nocheck > $O/gemm_lab.txt 2>&1
timeout 600 python bench.py --steps 150 --warmup 15 --no-cpu-baseline 2>$O/bench_base.err | tail -1 > $O/bench_base.json
MMT_TILE_192=1 timeout 600 python bench.py --steps 150 --warmup 15 --no-cpu-baseline 2>$O/bench_192.err | tail -1 > $O/bench_192.json
MMT_TILE_N3072=15 timeout 600 python bench.py --steps 150 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_3072only.json
cd /tmp
rm -rf /tmp/prof && MMT_TILE_192=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o step -- python $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline > $O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB --by-grid --top 100 > $O/kernel_stats_by_grid.txt 2>&1
python $R/tools/rocpd_stats.py $DB --sequence 420 > $O/kernel_sequence.txt 2>&1
cd $R
timeout 300 python tools/torch_glue_profile.py > $O/glue.txt 2>&1
ls -la $O
