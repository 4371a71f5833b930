# Round 6: riders with ONE reporting block per hosting launch (instead of an atomic from every block with a tile).
source "$(dirname "$0")/r06_common.sh"
cd $R
timeout 600 python -m pytest tests/test_optim_gpu.py -x -q -k "queue or riders" > $O/pytest_sel.txt 2>&1; tail -2 $O/pytest_sel.txt
S="--steps 200 --warmup 20"
for i in 1 2; do
  ab serial_$i "$S" "MMT_X=0"
  ab adaptive8_$i "$S --adam-riders" "MMT_RIDER_CAP=8"
  ab adaptive24_$i "$S --adam-riders" "MMT_RIDER_CAP=24"
  ab adaptive64_$i "$S --adam-riders" "MMT_X=0"
  ab p1c64_$i "$S --adam-riders" "MMT_RIDER_CAP=64 MMT_RIDER_PASSES=1"
  ab p2c24_$i "$S --adam-riders" "MMT_RIDER_CAP=24 MMT_RIDER_PASSES=2"
done
T="--text-tower native --steps 60 --warmup 10"
ab tower_serial "$T" "MMT_X=0"
ab tower_adaptive64 "$T --adam-riders" "MMT_X=0"
ab tower_p1c128 "$T --adam-riders" "MMT_RIDER_CAP=128 MMT_RIDER_PASSES=1"
prof packed_riders_sig "--steps 50 --warmup 10 --adam-riders" "MMT_RIDER_CAP=24"
