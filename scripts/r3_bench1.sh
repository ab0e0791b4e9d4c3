set -x
timeout -k 10 600 python -m pytest tests/test_gpu_modes.py tests/test_gpu_packed.py -m gpu -q -x > gpurun_out/r3_t11.log 2>&1; tail -3 gpurun_out/r3_t11.log
bash scripts/r3_exp.sh paired150 BWT_X=0
bash scripts/r3_exp.sh grch38 BWT_X=0
