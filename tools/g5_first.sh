# gemm5 (tile 24) first contact: parity, lab timings against tile 14, whole-step A/B
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wide_tiles and 24" 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
timeout 300 python tools/gemm_lab.py --tiles 14,24 --rows 3639 2>&1 | tee $O/lab_warm.txt
timeout 300 python tools/gemm_lab.py --tiles 14,24 --rows 3639 --instep --nocheck 2>&1 | tee $O/lab_instep.txt
for rep in 1 2; do
  for v in 0 24; do
    MMT_TILE_WIDE=$v timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-dense 2>$O/err_$v.log | tail -1 > $O/bench_${v}_$rep.json
    python -c "
import json; d = json.load(open('$O/bench_${v}_$rep.json')); print('MMT_TILE_WIDE=$v %.4f ms/step  %.0f pairs/s loss %s' % (d['ms_per_step'], d['value'], d.get('first_loss')))" | tee -a $O/summary.txt
  done
done
