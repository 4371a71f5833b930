# Round 6: weight gradients of short contractions (text tower) on the one-group kernel at two blocks per CU.
source "$(dirname "$0")/r06_common.sh"
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_text_bert_gpu.py -x -q -k "wgrad or text" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
T="--text-tower native --steps 60 --warmup 10"
for i in 1 2 3; do
  ab tower_phased_$i "$T" "MMT_WGRAD_SHORT_ROWS=0"
  ab tower_short_$i "$T" "MMT_X=0"
done
prof tower_short "--text-tower native --steps 30 --warmup 5"
grep -n "wgrad" $O/kernel_stats_tower_short_by_grid.txt | head
