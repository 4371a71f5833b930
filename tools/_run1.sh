cd /root/repo
timeout 900 python -m pytest tests/test_text_bert_gpu.py -x -q 2>&1 | tail -25 | cut -c1-400
