set -x
mkdir -p gpurun_out/parity_r3
python scripts/parity_at_scale.py --reads 4000000 > gpurun_out/parity_r3/parity_at_scale.json 2> gpurun_out/parity_r3/a.err || { tail -5 gpurun_out/parity_r3/a.err; }
python scripts/parity_at_scale.py --reads 3000000 --repeats 300 --max-subs 5 > gpurun_out/parity_r3/parity_at_scale_repeats.json 2> gpurun_out/parity_r3/b.err || { tail -5 gpurun_out/parity_r3/b.err; }
python scripts/parity_at_scale.py --reads 2000000 --length 150 > gpurun_out/parity_r3/parity_at_scale_150bp.json 2> gpurun_out/parity_r3/c.err || { tail -5 gpurun_out/parity_r3/c.err; }
cat gpurun_out/parity_r3/*.json | cut -c1-600
