cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad" 2>&1 | tail -2
for w in 0 1 0 1; do
MMT_WGRAD3=$w timeout 300 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=[x for x in d['roofline_top3'] if 'weight grad' in x['kernel']][0]
print('wgrad3 %s  %.4f ms/step  wgrad %.1f us frac %.3f' % ('$w', d['ms_per_step'], r['avg_launch_us'], r['frac']))"
done
rm -rf /tmp/pmc1; timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc1 -o p -- python $R/bench.py --config 4 --steps 4 --warmup 2 --no-cpu-baseline --no-dense > /dev/null 2>&1
python $R/tools/rocpd_pmc.py $(find /tmp/pmc1 -name "*.db") --top 12 2>&1 | grep -i "wgrad" | cut -c1-160
