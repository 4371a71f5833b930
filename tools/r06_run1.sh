# Round 6, first GPU contact of the Adam riders: new tests, same-box A/B of the step with / without riders and with
# MMT_TILE_PPN=6, the replayed sequence with riders, L2 / fabric counters of the replayed step, the full GPU suite.
#   gpurun --timeout 2400 -- 'bash tools/r06_run1.sh'
source "$(dirname "$0")/r06_common.sh"
cd $R
timeout 600 python -m pytest tests/test_optim_gpu.py -x -q > $O/pytest_optim.txt 2>&1; tail -3 $O/pytest_optim.txt
S="--steps 200 --warmup 20"
for i in 1 2; do
  ab riders_$i "$S" "MMT_X=0"
  ab serial_$i "$S --no-adam-riders" "MMT_X=0"
  ab riders_ppn6_$i "$S" "MMT_TILE_PPN=6"
  ab serial_ppn6_$i "$S --no-adam-riders" "MMT_TILE_PPN=6"
done
prof packed_riders "--steps 50 --warmup 10"
tail -3 $O/graph_sequence_packed_riders.txt
ab tower_riders "--text-tower native --steps 60 --warmup 10" "MMT_X=0"
ab tower_serial "--text-tower native --steps 60 --warmup 10 --no-adam-riders" "MMT_X=0"
pmc_cache cache_packed "--steps 12 --warmup 3 --no-adam-riders"
head -12 $O/pmc_cache_packed.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
