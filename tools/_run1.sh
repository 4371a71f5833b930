cd /root/repo
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["first_loss"], d["final_loss"], d["config"]["text_tower"])'
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
timeout 900 python -m pytest tests/test_text_bert_gpu.py -x -q 2>&1 | tail -3 | cut -c1-500
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
