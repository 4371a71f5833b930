# Round 6, riders with a fixed number of passes (no overshoot), whole backward as one range; cold-lab cache counters.
#   gpurun --timeout 2400 -- 'bash tools/r06_run3.sh'
source "$(dirname "$0")/r06_common.sh"
cd $R
timeout 600 python -m pytest tests/test_optim_gpu.py -x -q -k "queue or riders" > $O/pytest_sel.txt 2>&1; tail -2 $O/pytest_sel.txt
S="--steps 200 --warmup 20"
for i in 1 2; do
  ab serial_$i "$S --no-adam-riders" "MMT_X=0"
  ab adaptive24_$i "$S" "MMT_RIDER_CAP=24"
  ab p1c64_$i "$S" "MMT_RIDER_CAP=64 MMT_RIDER_PASSES=1"
  ab p2c24_$i "$S" "MMT_RIDER_CAP=24 MMT_RIDER_PASSES=2"
  ab p4c24_$i "$S" "MMT_RIDER_CAP=24 MMT_RIDER_PASSES=4"
  ab p2c64_$i "$S" "MMT_RIDER_CAP=64 MMT_RIDER_PASSES=2"
  ab serial_ppn6_$i "$S --no-adam-riders" "MMT_TILE_PPN=6"
done
T="--text-tower native --steps 60 --warmup 10"
for i in 1 2; do
  ab tower_serial_$i "$T --no-adam-riders" "MMT_X=0"
  ab tower_p1c64_$i "$T" "MMT_RIDER_CAP=64 MMT_RIDER_PASSES=1"
  ab tower_p1c128_$i "$T" "MMT_RIDER_CAP=128 MMT_RIDER_PASSES=1"
  ab tower_p1c256_$i "$T" "MMT_RIDER_CAP=256 MMT_RIDER_PASSES=1"
  ab tower_p2c128_$i "$T" "MMT_RIDER_CAP=128 MMT_RIDER_PASSES=2"
done
prof tower_p1c128 "--text-tower native --steps 30 --warmup 5" "MMT_RIDER_CAP=128 MMT_RIDER_PASSES=1"
prof packed_p2c24 "--steps 50 --warmup 10" "MMT_RIDER_CAP=24 MMT_RIDER_PASSES=2"
# the FFN-down GEMM in the lab, one process per cache state, same counters as pmc_cache of the step (r06_run1)
for mode in warm cold producer; do
  for grp in "h:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "r:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum" "w:TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum"; do
    tag=${grp%%:*}; ctr=${grp#*:}
    rm -rf /tmp/pmc_lab_${mode}_$tag && (cd /tmp && timeout 300 rocprofv3 --pmc $ctr -d /tmp/pmc_lab_${mode}_$tag -o p -- python $R/tools/cold_lab.py --pmc $mode > /dev/null 2>&1)
  done
  python $R/tools/rocpd_pmc.py $(find /tmp/pmc_lab_${mode}_h /tmp/pmc_lab_${mode}_r /tmp/pmc_lab_${mode}_w -name "*.db") --csv $O/pmc_cache_lab_$mode.csv --match gemm2 --top 6 > $O/pmc_cache_lab_$mode.txt 2>&1
  head -4 $O/pmc_cache_lab_$mode.txt
done
