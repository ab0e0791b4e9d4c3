set -x
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3_t10.log 2>&1; tail -4 gpurun_out/r3_t10.log
bash scripts/r3_exp.sh grch38-repeats BWT_X=0
bash scripts/r3_exp.sh grch38 BWT_X=0
bash scripts/r3_exp.sh paired150 BWT_X=0
