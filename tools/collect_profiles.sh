# gpurun_out/r04_final/* (scratch, merged back by gpurun) -> profiles/r04_* (tracked)
set -e
cd "$(dirname "$0")/.."
for f in gpurun_out/r04_final/*.json gpurun_out/r04_final/*.txt gpurun_out/r04_final/*.csv; do
  b=$(basename $f)
  case $b in
    pmc_*) cp $f profiles/r04_$b ;;
    gemm2_budget.txt) cp $f profiles/r04_gemm2_budget.txt ;;
    attn_budget.txt) cp $f profiles/r04_attn_budget.txt ;;
    *) cp $f profiles/r04_final_$b ;;
  esac
done
ls profiles | grep r04
