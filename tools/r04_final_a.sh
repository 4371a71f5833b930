# Round-4 artifacts, part A: the GPU test suite and the smoke test at HEAD -> gpurun_out/r04_final/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_final
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 > $O/pytest_gpu.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -3 > $O/smoke.txt
cat $O/pytest_gpu.txt $O/smoke.txt
