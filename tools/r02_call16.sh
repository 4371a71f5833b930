cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c16
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_text_bert_gpu.py tests/test_ragged_gpu.py -q -m gpu -x 2>&1 | tail -15 > $O/pytest_gpu.txt
