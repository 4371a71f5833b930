# gpurun_out/final/* (scratch, merged back by gpurun) -> profiles/r03_* (tracked)
set -e
cd "$(dirname "$0")/.."
for f in gpurun_out/final/*.json gpurun_out/final/*.txt gpurun_out/final/*.csv; do
  b=$(basename $f)
  case $b in
    pmc_kernels.csv) cp $f profiles/r03_pmc_kernels.csv ;;
    pmc_kernels.txt) cp $f profiles/r03_pmc_kernels.txt ;;
    pmc_*) cp $f profiles/r03_$b ;;
    fork_lab.txt) cp $f profiles/r03_fork_lab_final.txt ;;
    *) cp $f profiles/r03_final_$b ;;
  esac
done
ls profiles | grep r03
