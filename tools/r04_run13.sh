cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 800 python -m pytest tests/test_dp_gpu.py -q -x -k "input_slots" 2>&1 | grep -v "socket.cpp\|Gloo" | tail -6
for i in 1 2; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('resident   %.4f ms/step' % d['ms_per_step'])"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense --host-inputs 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('host-inputs %.4f ms/step | %s' % (d['ms_per_step'], d['config']['inputs']))"
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense --host-inputs --ragged-inputs 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('host ragged %.4f ms/step | %s' % (d['ms_per_step'], d['config']['inputs']))"
