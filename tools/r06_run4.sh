# Round 6: text tower with the slab-reading LayerNorms of the short-batch split-K GEMMs (+ dQKV), weight gradients on 256x256
# tiles for short batches; the full GPU suite on the new defaults (riders off, live-row hints).
#   gpurun --timeout 2400 -- 'bash tools/r06_run4.sh'
source "$(dirname "$0")/r06_common.sh"
cd $R
timeout 900 python -m pytest tests/test_text_bert_gpu.py tests/test_cenet_gpu.py -x -q -k "text or hint or configA" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
T="--text-tower native --steps 60 --warmup 10"
for i in 1 2; do
  ab tower_r05_$i "$T" "MMT_SPLITK_LN=0"
  ab tower_slabln_$i "$T" "MMT_X=0"
  ab tower_slabln_w3_$i "$T" "MMT_WGRAD3_ROWS=256 MMT_WGRAD3_TILES=96"
done
prof tower "--text-tower native --steps 30 --warmup 5"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
