# Round-5 evidence, part 1b: configs[3] / configs[4] again (their N = hidden GEMMs carry the long-K grid tag now)
source "$(dirname "$0")/final_common.sh"
cd $R
for c in ${1:-3 4}; do
  if [ $c = 3 ]; then a="--config 3 --steps 30 --warmup 5"; b="--config 3 --steps 6 --warmup 2"; k="--config 3 --steps 60 --warmup 10"; fi
  if [ $c = 4 ]; then a="--config 4 --steps 15 --warmup 3"; b="--config 4 --steps 4 --warmup 2"; k="--config 4 --steps 30 --warmup 5"; fi
  prof config$c "$a"
  pmc config$c "$b"
  cd $R
  timeout 600 python bench.py $k 2>/dev/null | tail -1 > $O/bench_config$c.json
  line $O/bench_config$c.json
done
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_large_sim_gpu.py -q -x -k "eight_phase or large or row_block" 2>&1 | tail -2
