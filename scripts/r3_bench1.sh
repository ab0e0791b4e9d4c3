set -x
timeout -k 10 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py tests/test_gpu_modes.py -m gpu -q -x > gpurun_out/r3_t9.log 2>&1; tail -4 gpurun_out/r3_t9.log
bash scripts/r3_exp.sh grch38-repeats BWT_X=0
bash scripts/r3_exp.sh paired150 BWT_X=0
