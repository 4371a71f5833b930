# Round-5 evidence, part 2: the bench JSON lines (they read profiles/r05_pmc_*.csv and r05_kernel_stats_*_by_grid.csv: run part 1
# first and commit / keep its files in profiles/).   gpurun -- 'bash tools/final_2.sh'
source "$(dirname "$0")/final_common.sh"
cd $R
timeout 900 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench_packed.json
timeout 600 python bench.py --steps 200 --warmup 20 --dense --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_dense.json
timeout 600 python bench.py --config 3 --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bench_config3.json
timeout 600 python bench.py --config 4 --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_config4.json
timeout 600 python bench.py --steps 100 --warmup 10 --text-tower native --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_native_text_tower.json
for i in 1 2; do
  timeout 600 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_resident_$i.json
  timeout 600 python bench.py --steps 300 --warmup 20 --host-inputs --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_host_inputs_$i.json
  timeout 600 python bench.py --steps 300 --warmup 20 --host-inputs --ragged-inputs --no-cpu-baseline --no-dense 2>/dev/null | tail -1 > $O/bench_host_inputs_ragged_$i.json
done
# N > 1 control flow on one GPU: gloo ranks sharing the device (functional smoke -- the numbers are gloo through host memory)
MMT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --grad-algo rs_ag --shard-optimizer 2>/dev/null | tail -1 > $O/bench_2ranks_gloo_sharded_adam.json
MMT_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_8ranks_gloo_allreduce.json
MMT_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 10 --warmup 3 --no-cpu-baseline --grad-algo rs_ag --shard-optimizer 2>/dev/null | tail -1 > $O/bench_8ranks_gloo_sharded_adam.json
# same-box A/B of this round's switches (three alternations each)
for rep in 1 2 3; do
  for v in "MMT_TILE_PP=0" "MMT_TILE_PP=1" "MMT_TILE_PP=2" "MMT_FUSE_OUT_LN=1"; do
    env $v timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('%-20s rep $rep  %.4f ms/step  dense %.4f' % ('$v', d['ms_per_step'], d['dense']['ms_per_step']))" >> $O/switch_ab.txt
  done
done
for f in $O/bench_*.json; do line $f; done
cat $O/switch_ab.txt
