cd /root/repo
timeout 1500 python -m pytest tests/test_cenet_gpu.py tests/test_dp_gpu.py tests/test_text_bert_gpu.py -x -q 2>&1 | tail -3 | cut -c1-300
P='import json,sys
for l in sys.stdin:
  if l.startswith("{"):
    d=json.loads(l); print(d["ms_per_step"], d["config"]["text_tower"], d["final_loss"])'
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P"
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P"
