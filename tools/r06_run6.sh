# Round 6: six-deep ring (tile 26 / split-K) for launches that do not fill the chip: parity, tower and headline A/B.
source "$(dirname "$0")/r06_common.sh"
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_text_bert_gpu.py tests/test_dp_gpu.py -x -q -k "gemm or text or recaptures" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
T="--text-tower native --steps 60 --warmup 10"
S="--steps 200 --warmup 20"
for i in 1 2; do
  ab tower_ring3_$i "$T" "MMT_DEEP_RING=0"
  ab tower_ring6_$i "$T" "MMT_X=0"
  ab head_ring3_$i "$S" "MMT_DEEP_RING=0"
  ab head_ring6_$i "$S" "MMT_X=0"
done
prof tower_ring6 "--text-tower native --steps 30 --warmup 5"
