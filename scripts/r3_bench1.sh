set -x
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_t6.log 2>&1; tail -8 gpurun_out/r3_t6.log
timeout -k 10 600 python bench.py --workload grch38-repeats --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_rep2.json 2> gpurun_out/r3_rep2.err || { tail -30 gpurun_out/r3_rep2.err; exit 1; }
