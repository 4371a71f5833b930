# gpurun_out/final/* (scratch, merged back by gpurun) -> profiles/r05_* (tracked)
set -e
cd "$(dirname "$0")/.."
for f in gpurun_out/final/*.json gpurun_out/final/*.txt gpurun_out/final/*.csv; do
  [ -f "$f" ] || continue
  b=$(basename $f)
  case $b in
    pmc_*|kernel_stats_*_by_grid.csv) cp $f profiles/r05_$b ;;
    gemm2_budget.txt|g5_budget_*.txt) cp $f profiles/r05_$b ;;
    *) cp $f profiles/r05_final_$b ;;
  esac
done
ls profiles | grep r05
