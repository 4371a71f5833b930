cd /root/repo
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | cut -c1-300
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["first_loss"], d["final_loss"], d["config"]["text_tower"])'
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --text-tower native 2>&1 | tail -1 | python -c "$P"
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
