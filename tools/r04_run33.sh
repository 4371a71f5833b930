cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad" 2>&1 | tail -2 | cut -c1-200
MMT_HIP_LIB=mmt_amd/lib/libmmt_hip.so timeout 200 python tools/wgrad3_lab.py 2>&1 | tail -1
MMT_WGRAD3=0 timeout 200 python tools/wgrad3_lab.py 2>&1 | tail -1
timeout 300 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=[x for x in d['roofline_top3'] if 'weight grad' in x['kernel']][0]
print('config 4  %.4f ms/step  wgrad %.1f us frac %.3f  exec %.3f' % (d['ms_per_step'], r['avg_launch_us'], r['frac'], d['executed_mfma_frac']))"
