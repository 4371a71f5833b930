# gpurun_out/final/* (scratch, merged back by gpurun) -> profiles/r02_final_* (tracked)
set -e
cd "$(dirname "$0")/.."
for f in gpurun_out/final/*.json gpurun_out/final/*.txt gpurun_out/final/*.csv; do
  b=$(basename $f)
  case $b in
    pmc_kernels.csv) cp $f profiles/r02_pmc_kernels.csv ;;
    pmc_kernels.txt) cp $f profiles/r02_pmc_kernels.txt ;;
    *) cp $f profiles/r02_final_$b ;;
  esac
done
ls profiles | grep r02
