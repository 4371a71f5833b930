cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
b() { timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dense $2 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('%-28s %.4f ms/step  slots %d' % ('$1', d['ms_per_step'], d['config']['input_slots']))"; }
for i in 1 2; do
b resident ""
b host-inputs "--host-inputs"
b host-inputs-streamwait "--host-inputs --stream-wait-uploads"
b host-ragged "--host-inputs --ragged-inputs"
b host-inputs-8slots "--host-inputs --input-slots 8"
done
