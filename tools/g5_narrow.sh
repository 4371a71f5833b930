# gemm5 on the long-K GEMMs with narrow outputs: parity, lab budgets (warm / cold), then the step with the policy switch.
#   gpurun -- 'bash tools/g5_narrow.sh'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g5n
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "wide_tiles or persistent_gemm" 2>&1 | tail -5 | tee $O/pytest.txt
MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/g5_budget.py 3639 --narrow 2>&1 | grep -v amdgpu.ids | tee $O/budget_warm.txt
MMT_HIP_LIB=$R/mmt_amd/lib/libmmt_hip_instr.so timeout 300 python tools/g5_budget.py 3639 --narrow --cold 2>&1 | grep -v amdgpu.ids | tee $O/budget_cold.txt
for rep in 1 2; do
  for v in 0 1 4; do
    MMT_TILE_PPN=$v timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>$O/err_$v.log | tail -1 > $O/bench_${v}_$rep.json
    python -c "
import json; d = json.load(open('$O/bench_${v}_$rep.json')); print('MMT_TILE_PPN=$v %.4f ms/step  %.0f pairs/s  dense %.4f  loss %s' % (d['ms_per_step'], d['value'], d['dense']['ms_per_step'] if d.get('dense') else 0, d.get('first_loss')))" | tee -a $O/summary.txt
  done
done
