cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_lab3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_eval_loop_gpu.py tests/test_large_sim_gpu.py tests/test_cenet_gpu.py tests/test_dp_gpu.py -q -m gpu -s -k "eval_loop or separated or row_block or bench_configuration or reduce_scatter or staged or rccl" 2>&1 | grep -v "^$" | tail -30 > $O/pytest_new.txt
cat $O/pytest_new.txt
for rep in 1 2; do
for sp in 1 0; do
  MMT_ATTN_BWD_SPLIT=$sp timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-dense 2>$O/err_ab.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); print('attn bwd split=$sp  %.4f ms/step  %.0f pairs/s' % (d['ms_per_step'], d['value']))
" >> $O/ab.txt
done
done
cat $O/ab.txt
timeout 300 python bench.py --config 3 --steps 60 --warmup 10 2>$O/err_c3.log | tail -1 > $O/bench_config3.json
timeout 600 python bench.py --config 4 --steps 30 --warmup 5 2>$O/err_c4.log | tail -1 > $O/bench_config4.json
python - <<'PY'
import json
for n in (3, 4):
  try:
    d = json.load(open('gpurun_out/r03_lab3/bench_config%d.json' % n))
    print('config', n, '%.3f ms/step' % d['ms_per_step'], '%.0f pairs/s' % d['value'], 'dense', d['dense'], 'enc dense frac %.3f exec frac %.3f' % (d['encoder_dense_mfma_frac'], d['executed_mfma_frac']))
    for r in d.get('roofline_top3', []):
      print('   ', r['kernel'][:50], '%.1f us' % r['avg_launch_us'], 'frac %.3f' % r['frac'])
    if 'similarity_loss_row_block' in d: print('   row block', d['similarity_loss_row_block'])
  except Exception as e:
    print('config', n, 'failed', e)
PY
MMT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gloo2.txt 2>$O/err_gloo2.log
tail -1 $O/bench_gloo2.txt | cut -c1-600
