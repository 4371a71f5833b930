cd /root/repo
timeout 900 python -m pytest tests/test_optim_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -8 | cut -c1-300
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["final_loss"])'
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
