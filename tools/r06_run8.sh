# Round 6: the whole GPU suite with the optimizer riders ON in every single-rank GraphedTrainStep (MMT_ADAM_RIDERS=1): the opt-in
# path against every parity / trainer-loop / checkpoint test.   gpurun --timeout 1800 -- 'bash tools/r06_run8.sh'
source "$(dirname "$0")/r06_common.sh"
cd $R
MMT_ADAM_RIDERS=1 timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_riders_on.txt 2>&1; tail -4 $O/pytest_gpu_riders_on.txt
