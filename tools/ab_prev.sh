# same-box A/B of whole steps: the tree in ab_prev/ (an older commit, built there) against the working tree, alternating
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="${@:---steps 200 --warmup 20 --no-cpu-baseline --no-dense}"
for i in 1 2 3; do
  for t in ab_prev .; do
    (cd $R/$t && timeout 300 python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('%-8s %.4f ms/step  %.0f pairs/s' % ('$t', d['ms_per_step'], d['value']))")
  done
done
