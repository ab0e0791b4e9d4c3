set -x
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_b4.json 2> gpurun_out/r3_b4.err || { tail -30 gpurun_out/r3_b4.err; exit 1; }
