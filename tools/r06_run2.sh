# Round 6, second contact of the Adam riders (fetch-add claims, capped riders, K-loop-end signal): new tests, same-box A/B of
# rider caps against the serial optimizer, sequences, text tower.   gpurun --timeout 2400 -- 'bash tools/r06_run2.sh'
source "$(dirname "$0")/r06_common.sh"
cd $R
timeout 600 python -m pytest tests/test_optim_gpu.py tests/test_cenet_gpu.py -x -q -k "queue or riders or hint or configB" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
S="--steps 200 --warmup 20"
for i in 1 2; do
  ab serial_$i "$S --no-adam-riders" "MMT_X=0"
  ab riders64_$i "$S" "MMT_X=0"
  ab riders24_$i "$S" "MMT_RIDER_CAP=24"
  ab riders128_$i "$S" "MMT_RIDER_CAP=128"
done
prof packed_riders "--steps 50 --warmup 10"
tail -3 $O/graph_sequence_packed_riders.txt
ab tower_serial "--text-tower native --steps 60 --warmup 10 --no-adam-riders" "MMT_X=0"
ab tower_riders "--text-tower native --steps 60 --warmup 10" "MMT_X=0"
ab tower_riders128 "--text-tower native --steps 60 --warmup 10" "MMT_RIDER_CAP=128"
prof tower_riders "--text-tower native --steps 30 --warmup 5"
prof tower_serial "--text-tower native --steps 30 --warmup 5 --no-adam-riders"
