cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_lab7
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_cenet_gpu.py -q -m gpu -x -k "wgrad or every_parameter or matches_reference" 2>&1 | grep -v "^$" | cut -c1-300 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
for ls in 1 0; do
  MMT_WGRAD_LOCKSTEP=$ls timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-dense 2>$O/err_ab.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); print('wgrad lockstep=$ls  %.4f ms/step  %.0f pairs/s  wgrad %.1f us frac %.3f' % (d['ms_per_step'], d['value'], [r for r in d['roofline_top3'] if 'weight grad' in r['kernel']][0]['avg_launch_us'], [r for r in d['roofline_top3'] if 'weight grad' in r['kernel']][0]['frac']))
"
done
done
for ls in 1 0; do
  MMT_WGRAD_LOCKSTEP=$ls timeout 300 python bench.py --config 4 --steps 30 --warmup 5 --no-dense 2>$O/err_ab.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); print('config 4 wgrad lockstep=$ls  %.4f ms/step  wgrad %.1f us frac %.3f' % (d['ms_per_step'], [r for r in d['roofline_top3'] if 'weight grad' in r['kernel']][0]['avg_launch_us'], [r for r in d['roofline_top3'] if 'weight grad' in r['kernel']][0]['frac']))
"
done
python -m mmt_amd.build --instr > /dev/null 2>&1; MMT_HIP_LIB=mmt_amd/lib/libmmt_hip_instr.so python tools/wgrad_instr.py 2>&1 | tail -6
