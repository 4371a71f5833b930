cd /root/repo
P='import json,sys
for l in sys.stdin:
  if l.startswith("{"):
    d=json.loads(l); print(d["ms_per_step"], d["config"]["text_tower"], d["final_loss"])'
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P"
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P"
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --text-tower native 2>/dev/null | python -c "$P"
