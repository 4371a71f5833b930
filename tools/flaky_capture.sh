# Repeats tests/test_dp_gpu.py to catch the captured-collectives test's intermittent failure with its full message.
#   gpurun -- 'bash tools/flaky_capture.sh 2'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/flaky
for i in $(seq 1 ${1:-2}); do
  timeout 600 python -m pytest tests/test_dp_gpu.py -q -m gpu -x -k "not bench_two_ranks" > gpurun_out/flaky/file_$i.txt 2>&1
  rc=$?
  echo "run $i rc=$rc $(tail -1 gpurun_out/flaky/file_$i.txt)"
  if [ $rc -ne 0 ]; then grep -n "Error\|error\|what()\|terminate" gpurun_out/flaky/file_$i.txt | cut -c1-500 | head -40; break; fi
done
