cd /root/repo
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_cenet_gpu.py -x -q 2>&1 | tail -3 | cut -c1-300
P='import json,sys
for l in sys.stdin:
  if l.startswith("{"):
    d=json.loads(l); print(d["ms_per_step"], d["config"]["grad_sync"], d["final_loss"])'
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P"
for gs in single staged; do echo forced $gs; timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --force-collectives --grad-sync $gs 2>/dev/null | python -c "$P"; done
export MMT_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 60 --warmup 10 2>/dev/null | python -c "$P"
