cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_text_heads_gpu.py tests/test_optim_gpu.py -q -x 2>&1 | tail -30 > $O/pytest_dp.txt
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_dp_gpu.py 2>&1 | tail -15 > $O/pytest_gpu.txt
timeout 900 python bench.py --steps 150 --warmup 15 2>$O/bench.err | tail -1 > $O/bench.json
MMT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 20 --warmup 5 --grad-dtype bf16 2>$O/bench2.err | tail -1 > $O/bench_2rank_gloo_bf16.json
ls $O
