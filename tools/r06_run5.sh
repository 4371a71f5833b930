# Round 6: configs[4] with the K = 1024 GEMMs on the persistent 128x128 kernel (tile 24) instead of the 256x256 one (tile 21);
# fill sweep with the host's live-row counts.   gpurun --timeout 2400 -- 'bash tools/r06_run5.sh'
source "$(dirname "$0")/r06_common.sh"
cd $R
C="--config 4 --steps 12 --warmup 3"
for i in 1 2; do
  ab cfg4_t21_$i "$C" "MMT_X=0"
  ab cfg4_t24_$i "$C" "MMT_TILE_BIG_KMIN=2048"
done
prof cfg4_t24 "--config 4 --steps 8 --warmup 2" "MMT_TILE_BIG_KMIN=2048"
C3="--config 3 --steps 30 --warmup 5"
ab cfg3_1 "$C3" "MMT_X=0"
for f in 0.25 0.51 0.75 1.0; do
  ab fill_${f}_auto "--fill $f --steps 100 --warmup 10" "MMT_X=0"
  ab fill_${f}_n18 "--fill $f --steps 100 --warmup 10" "MMT_TILE_NARROW=18 MMT_LIVE_FRACTION=0.3"
  ab fill_${f}_n13 "--fill $f --steps 100 --warmup 10" "MMT_TILE_NARROW=13 MMT_TILE_PPN=0"
  ab fill_${f}_n24 "--fill $f --steps 100 --warmup 10" "MMT_LIVE_FRACTION=1.0"
done
