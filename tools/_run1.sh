cd /root/repo
timeout 900 python -m pytest tests/test_text_bert_gpu.py -x -q 2>&1 | tail -12 | cut -c1-500
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["first_loss"], d["final_loss"], d["config"]["text_tower"])'
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --text-tower native 2>&1 | tail -1 | python -c "$P"
