cd /root/repo
python tools/attn_lab.py 2>&1 | tail -12
