# Lab: tile 25 (gemm5 on 128x64 tiles) on the packed K = hidden GEMMs with N = 512.   gpurun -- 'bash tools/ab_ppn2.sh'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_ppn2
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "wide_tiles or persistent_gemm" 2>&1 | tail -2
for rep in 1 2 3; do
  for v in 1 6 5; do
    MMT_TILE_PPN=$v timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-dense 2>$O/err.log | tail -1 > $O/b.json
    python -c "
import json; d = json.load(open('$O/b.json')); print('MMT_TILE_PPN=$v packed %.4f ms/step  loss %s' % (d['ms_per_step'], d.get('first_loss')))" | tee -a $O/summary.txt
  done
done
