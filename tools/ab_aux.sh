# Lab: cache-policy bits on gemm5's LDS-DMA requests (lab libraries built with MMT_LAB_DEFINES="G5_LAB_AUX_A=.. G5_LAB_AUX_B=..").
#   gpurun -- 'bash tools/ab_aux.sh'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_aux
mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in libmmt_hip.so libmmt_hip_lab_aux_2_0.so libmmt_hip_lab_aux_2_2.so libmmt_hip_lab_aux_17_0.so; do
    MMT_HIP_LIB=$R/mmt_amd/lib/$lib MMT_TILE_PPN=4 timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>$O/err.log | tail -1 > $O/b.json
    python -c "
import json; d = json.load(open('$O/b.json')); print('%-32s PPN=4 packed %.4f  unpacked %.4f ms/step  loss %s' % ('$lib', d['ms_per_step'], d['dense']['ms_per_step'], d.get('first_loss')))" | tee -a $O/summary.txt
  done
done
